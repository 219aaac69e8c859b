/* rans_oracle.c - TEST INFRASTRUCTURE ONLY (parity checker; never linked into the product).
 *
 * Plain-C, single-threaded restatement of the reference's host entropy coder:
 *   byte-wise rANS primitives        src/cpp/py_rans/rans_byte.h:61-141  (ryg_rans, 32-bit state,
 *                                    SCALE_BITS 16, L = 1<<23)
 *   escape / bypass coding           src/cpp/py_rans/rans.cpp:28-58,95-140,356-395
 *   reverse-order task flush         src/cpp/py_rans/rans.cpp:202-243
 *   y / z symbol -> cdf selection    src/cpp/py_rans/rans.cpp:165-200,397-429
 *   two-coder split and stream merge src/cpp/py_rans/py_rans.cpp:20-67,109-151,175-262
 *   pmf_to_quantized_cdf             src/cpp/py_rans/py_rans.cpp:307-364
 * Pinned bit-exactly against the reference's own coder (oracle/_ref, built from the reference
 * sources in place) by tests/test_oracle_rans.py and by the committed KAT fixtures.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define SCALE_BITS 16
#define RANS_L (1u << 23)
#define RENORM_SHIFT (23 - SCALE_BITS + 8)
#define BYPASS_BITS 2
#define BYPASS_MAX ((1 << BYPASS_BITS) - 1)
#define MAX_GROUPS 8
#define MAX_TASKS 64

typedef struct {
    int n, stride;
    int32_t* cdf; /* [n][stride] */
    int32_t* sizes;
    int32_t* offsets;
} Group;

typedef struct {
    int is_z;
    int n;
    int16_t* y;
    int8_t* z;
    int group, start_offset, per_channel;
} Task;

typedef struct {
    Task tasks[MAX_TASKS];
    int n_tasks;
    uint8_t* stream;
    int stream_len;
} Enc;

typedef struct {
    uint8_t* buf;
    int len;
    int pos;
    uint32_t state;
} Dec;

typedef struct OrcCoder {
    Group groups[MAX_GROUPS];
    int n_groups;
    int enc_two, dec_two;
    Enc enc[2];
    uint8_t* merged;
    int merged_len;
    Dec dec[2];
} OrcCoder;

OrcCoder* orc_coder_new(void) { return (OrcCoder*)calloc(1, sizeof(OrcCoder)); }

static void enc_clear(Enc* e)
{
    for (int i = 0; i < e->n_tasks; ++i) {
        free(e->tasks[i].y);
        free(e->tasks[i].z);
    }
    e->n_tasks = 0;
    free(e->stream);
    e->stream = NULL;
    e->stream_len = 0;
}

void orc_coder_free(OrcCoder* c)
{
    if (!c) return;
    for (int g = 0; g < c->n_groups; ++g) {
        free(c->groups[g].cdf);
        free(c->groups[g].sizes);
        free(c->groups[g].offsets);
    }
    enc_clear(&c->enc[0]);
    enc_clear(&c->enc[1]);
    free(c->merged);
    free(c->dec[0].buf);
    free(c->dec[1].buf);
    free(c);
}

int orc_add_cdf(OrcCoder* c, const int32_t* cdf, int n, int stride, const int32_t* sizes,
                const int32_t* offsets)
{
    if (c->n_groups >= MAX_GROUPS) return -1;
    Group* g = &c->groups[c->n_groups];
    g->n = n;
    g->stride = stride;
    g->cdf = (int32_t*)malloc(sizeof(int32_t) * (size_t)n * stride);
    g->sizes = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
    g->offsets = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
    memcpy(g->cdf, cdf, sizeof(int32_t) * (size_t)n * stride);
    memcpy(g->sizes, sizes, sizeof(int32_t) * (size_t)n);
    memcpy(g->offsets, offsets, sizeof(int32_t) * (size_t)n);
    return c->n_groups++;
}

/* ---------------------------------------------------------------- encoder */

static inline void put_bits(uint32_t* r, uint8_t** ptr, uint32_t val)
{
    const uint32_t freq = 1u << (SCALE_BITS - BYPASS_BITS);
    const uint32_t x_max = freq << RENORM_SHIFT;
    while (*r >= x_max) {
        *(--(*ptr)) = (uint8_t)(*r & 0xff);
        *r >>= 8;
    }
    *r = (*r << BYPASS_BITS) | val;
}

static inline void put_sym(uint32_t* r, uint8_t** ptr, uint32_t start, uint32_t freq)
{
    const uint32_t x_max = freq << RENORM_SHIFT;
    while (*r >= x_max) {
        *(--(*ptr)) = (uint8_t)(*r & 0xff);
        *r >>= 8;
    }
    *r = ((*r / freq) << SCALE_BITS) + (*r % freq) + start;
}

static void encode_one(uint32_t* r, uint8_t** ptr, int32_t symbol, const Group* g, int cdf_idx)
{
    const int32_t* cdf = g->cdf + (size_t)cdf_idx * g->stride;
    const int32_t max_value = g->sizes[cdf_idx] - 2;
    int32_t value = symbol - g->offsets[cdf_idx];
    uint32_t raw = 0;
    if (value < 0) {
        raw = (uint32_t)(-2 * value - 1);
        value = max_value;
    } else if (value >= max_value) {
        raw = (uint32_t)(2 * (value - max_value));
        value = max_value;
    }
    if (value == max_value) {
        uint16_t bins[40];
        int nb = 0;
        int32_t n_bypass = 0;
        while ((raw >> (n_bypass * BYPASS_BITS)) != 0) ++n_bypass;
        int32_t v = n_bypass;
        while (v >= BYPASS_MAX) {
            bins[nb++] = BYPASS_MAX;
            v -= BYPASS_MAX;
        }
        bins[nb++] = (uint16_t)v;
        for (int32_t j = 0; j < n_bypass; ++j) bins[nb++] = (uint16_t)((raw >> (j * BYPASS_BITS)) & BYPASS_MAX);
        for (int i = nb - 1; i >= 0; --i) put_bits(r, ptr, bins[i]);
    }
    put_sym(r, ptr, (uint32_t)cdf[value], (uint32_t)(cdf[value + 1] - cdf[value]));
}

static void enc_flush(OrcCoder* c, Enc* e)
{
    int64_t total = 0;
    for (int i = 0; i < e->n_tasks; ++i) total += e->tasks[i].n;
    free(e->stream);
    e->stream = NULL;
    e->stream_len = 0;
    if (total == 0) return;
    /* the reference sizes its scratch as one byte per symbol (rans.cpp:221); escape-heavy
     * inputs can exceed that there (undefined behaviour) - the oracle just allocates more. */
    const int64_t cap = total * 8 + 64;
    uint8_t* buf = (uint8_t*)malloc((size_t)cap);
    uint8_t* end = buf + cap;
    uint8_t* ptr = end;
    uint32_t r = RANS_L;
    for (int ti = e->n_tasks - 1; ti >= 0; --ti) {
        const Task* t = &e->tasks[ti];
        const Group* g = &c->groups[t->group];
        if (t->is_z) {
            for (int i = t->n - 1; i >= 0; --i)
                encode_one(&r, &ptr, t->z[i], g, i / t->per_channel + t->start_offset);
        } else {
            for (int i = t->n - 1; i >= 0; --i) {
                const int32_t cs = t->y[i];
                encode_one(&r, &ptr, cs >> 8, g, cs & 0xff);
            }
        }
    }
    ptr -= 4;
    ptr[0] = (uint8_t)(r >> 0);
    ptr[1] = (uint8_t)(r >> 8);
    ptr[2] = (uint8_t)(r >> 16);
    ptr[3] = (uint8_t)(r >> 24);
    e->stream_len = (int)(end - ptr);
    e->stream = (uint8_t*)malloc((size_t)e->stream_len);
    memcpy(e->stream, ptr, (size_t)e->stream_len);
    free(buf);
}

void orc_enc_reset(OrcCoder* c)
{
    enc_clear(&c->enc[0]);
    enc_clear(&c->enc[1]);
}

void orc_enc_set_two(OrcCoder* c, int two) { c->enc_two = two; }

static void push_y(Enc* e, const int16_t* s, int n, int group)
{
    Task* t = &e->tasks[e->n_tasks++];
    memset(t, 0, sizeof(*t));
    t->n = n;
    t->y = (int16_t*)malloc(sizeof(int16_t) * (size_t)(n > 0 ? n : 1));
    memcpy(t->y, s, sizeof(int16_t) * (size_t)n);
    t->group = group;
}

static void push_z(Enc* e, const int8_t* s, int n, int group, int start, int per_channel)
{
    Task* t = &e->tasks[e->n_tasks++];
    memset(t, 0, sizeof(*t));
    t->is_z = 1;
    t->n = n;
    t->z = (int8_t*)malloc((size_t)(n > 0 ? n : 1));
    memcpy(t->z, s, (size_t)n);
    t->group = group;
    t->start_offset = start;
    t->per_channel = per_channel;
}

void orc_enc_y(OrcCoder* c, const int16_t* s, int n, int group)
{
    if (c->enc_two) {
        const int n0 = n / 2;
        push_y(&c->enc[0], s, n0, group);
        push_y(&c->enc[1], s + n0, n - n0, group);
    } else {
        push_y(&c->enc[0], s, n, group);
    }
}

void orc_enc_z(OrcCoder* c, const int8_t* s, int n, int group, int start, int per_channel)
{
    if (c->enc_two) {
        const int n0 = n / 2;
        const int channel_half = n0 / per_channel;
        push_z(&c->enc[0], s, n0, group, start, per_channel);
        push_z(&c->enc[1], s + n0, n - n0, group, start + channel_half, per_channel);
    } else {
        push_z(&c->enc[0], s, n, group, start, per_channel);
    }
}

/* returns merged stream length */
int orc_enc_flush(OrcCoder* c)
{
    enc_flush(c, &c->enc[0]);
    free(c->merged);
    c->merged = NULL;
    c->merged_len = 0;
    if (!c->enc_two) {
        c->merged_len = c->enc[0].stream_len;
        c->merged = (uint8_t*)malloc((size_t)(c->merged_len > 0 ? c->merged_len : 1));
        if (c->merged_len) memcpy(c->merged, c->enc[0].stream, (size_t)c->merged_len);
        return c->merged_len;
    }
    enc_flush(c, &c->enc[1]);
    const uint8_t* s0 = c->enc[0].stream;
    const uint8_t* s1 = c->enc[1].stream;
    const int n0 = c->enc[0].stream_len, n1 = c->enc[1].stream_len;
    int identical = 0;
    int check = n0 < n1 ? n0 : n1;
    if (check > 8) check = 8;
    for (int i = 0; i < check; ++i) {
        if (s0[n0 - 1 - i] != 0) break;
        if (s1[n1 - 1 - i] != 0) break;
        ++identical;
    }
    if (identical == 0 && n0 > 0 && n1 > 0 && s0[n0 - 1] == s1[n1 - 1]) identical = 1;
    c->merged_len = n0 + n1 - identical;
    c->merged = (uint8_t*)malloc((size_t)(c->merged_len > 0 ? c->merged_len : 1));
    if (n0) memcpy(c->merged, s0, (size_t)n0);
    for (int i = 0; i < n1 - identical; ++i) c->merged[n0 + i] = s1[n1 - identical - 1 - i];
    return c->merged_len;
}

const uint8_t* orc_enc_stream(OrcCoder* c) { return c->merged; }

/* ---------------------------------------------------------------- decoder */

void orc_dec_set_two(OrcCoder* c, int two) { c->dec_two = two; }

static void dec_init(Dec* d, const uint8_t* s, int n, int reversed)
{
    free(d->buf);
    d->buf = (uint8_t*)malloc((size_t)n + 16);
    memset(d->buf, 0, (size_t)n + 16);
    if (reversed)
        for (int i = 0; i < n; ++i) d->buf[i] = s[n - 1 - i];
    else
        memcpy(d->buf, s, (size_t)n);
    d->len = n;
    d->pos = 0;
    d->state = 0;
    d->state = (uint32_t)d->buf[0] | ((uint32_t)d->buf[1] << 8) | ((uint32_t)d->buf[2] << 16) |
               ((uint32_t)d->buf[3] << 24);
    d->pos = 4;
}

void orc_dec_set_stream(OrcCoder* c, const uint8_t* s, int n)
{
    dec_init(&c->dec[0], s, n, 0);
    if (c->dec_two) dec_init(&c->dec[1], s, n, 1);
}

static inline uint32_t get_bits(Dec* d)
{
    const uint32_t val = d->state & ((1u << BYPASS_BITS) - 1);
    d->state >>= BYPASS_BITS;
    if (d->state < RANS_L) d->state = (d->state << 8) | d->buf[d->pos++];
    return val;
}

static int8_t decode_one(Dec* d, const Group* g, int cdf_idx)
{
    const int32_t* cdf = g->cdf + (size_t)cdf_idx * g->stride;
    const int32_t max_value = g->sizes[cdf_idx] - 2;
    const int32_t cum = (int32_t)(d->state & ((1u << SCALE_BITS) - 1));
    int s = 1;
    while (cdf[s++] <= cum) {
    }
    s -= 2;
    const uint32_t start = (uint32_t)cdf[s], freq = (uint32_t)(cdf[s + 1] - cdf[s]);
    d->state = freq * (d->state >> SCALE_BITS) + (d->state & ((1u << SCALE_BITS) - 1)) - start;
    while (d->state < RANS_L) d->state = (d->state << 8) | d->buf[d->pos++];
    int32_t value = s;
    if (value == max_value) {
        int32_t val = (int32_t)get_bits(d);
        int32_t n_bypass = val;
        while (val == BYPASS_MAX) {
            val = (int32_t)get_bits(d);
            n_bypass += val;
        }
        int32_t raw = 0;
        for (int j = 0; j < n_bypass; ++j) {
            val = (int32_t)get_bits(d);
            raw |= val << (j * BYPASS_BITS);
        }
        value = raw >> 1;
        if (raw & 1)
            value = -value - 1;
        else
            value += max_value;
    }
    return (int8_t)(value + g->offsets[cdf_idx]);
}

void orc_dec_y(OrcCoder* c, const uint8_t* idx, int n, int group, int8_t* out)
{
    const Group* g = &c->groups[group];
    if (c->dec_two) {
        const int n0 = n / 2;
        for (int i = 0; i < n0; ++i) out[i] = decode_one(&c->dec[0], g, idx[i]);
        for (int i = n0; i < n; ++i) out[i] = decode_one(&c->dec[1], g, idx[i]);
    } else {
        for (int i = 0; i < n; ++i) out[i] = decode_one(&c->dec[0], g, idx[i]);
    }
}

void orc_dec_z(OrcCoder* c, int total, int group, int start, int per_channel, int8_t* out)
{
    const Group* g = &c->groups[group];
    if (c->dec_two) {
        const int n0 = total / 2;
        const int channel_half = n0 / per_channel;
        for (int i = 0; i < n0; ++i) out[i] = decode_one(&c->dec[0], g, i / per_channel + start);
        for (int i = 0; i < total - n0; ++i)
            out[n0 + i] = decode_one(&c->dec[1], g, i / per_channel + start + channel_half);
    } else {
        for (int i = 0; i < total; ++i) out[i] = decode_one(&c->dec[0], g, i / per_channel + start);
    }
}

/* ---------------------------------------------------------------- cdf build */

/* py_rans.cpp:307-364.  out must hold n+1 entries.  Returns 0, or -1 if no frequency can be
 * stolen (the reference asserts). */
int orc_pmf_to_quantized_cdf(const float* pmf, int n, int precision, uint32_t* out)
{
    out[0] = 0;
    for (int i = 0; i < n; ++i) out[i + 1] = (uint32_t)(roundf(pmf[i] * (float)(1 << precision)) + 0.5);
    uint32_t total = 0;
    for (int i = 0; i <= n; ++i) total += out[i];
    for (int i = 0; i <= n; ++i) out[i] = (uint32_t)((((uint64_t)1 << precision) * out[i]) / total);
    for (int i = 1; i <= n; ++i) out[i] += out[i - 1];
    out[n] = 1u << precision;
    for (int i = 0; i < n; ++i) {
        if (out[i] == out[i + 1]) {
            uint32_t best_freq = ~0u;
            int best = -1;
            for (int j = 0; j < n; ++j) {
                const uint32_t f = out[j + 1] - out[j];
                if (f > 1 && f < best_freq) {
                    best_freq = f;
                    best = j;
                }
            }
            if (best < 0) return -1;
            if (best < i) {
                for (int j = best + 1; j <= i; ++j) out[j]--;
            } else {
                for (int j = i + 1; j <= best; ++j) out[j]++;
            }
        }
    }
    return 0;
}
