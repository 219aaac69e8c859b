// rans_host.cpp - host entropy coder of the DCVC-RT path (stays on the CPU by design,
// BASELINE.json north_star): byte-wise rANS with 32-bit state, 16-bit probabilities, escape
// ("bypass") coding of out-of-range symbols and the two-coder split of large frames.
//
// Own implementation; the byte stream is identical to the reference's
//   src/cpp/py_rans/rans_byte.h:61-141  (state, renormalisation, flush)
//   src/cpp/py_rans/rans.cpp:28-58,95-140,202-243,356-429  (bypass bits, task order, y/z cdf selection)
//   src/cpp/py_rans/py_rans.cpp:20-67,109-151,175-262     (split in two coders, stream merge)
// which tests/ verify against golden streams produced by the reference and against oracle/_ref.
// Differences that do not change the stream: per-symbol (start, freq) tables are flat arrays;
// the scratch buffer is sized for the worst case (the reference's one byte per symbol overflows on
// escape-heavy input); sentinel entries (low byte 0xFF) are dropped / zero-filled here instead of
// a boolean-mask compaction on the GPU.
#include <algorithm>
#if defined(__AVX2__)
#include <immintrin.h>
#endif
#include <cmath>
#include <new>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "common.hpp"

namespace {

constexpr int kScaleBits = 16;
constexpr uint32_t kRansL = 1u << 23;
constexpr int kRenormShift = 23 - kScaleBits + 8;
constexpr int kBypassBits = 2;
constexpr int kBypassMax = (1 << kBypassBits) - 1;
constexpr uint32_t kMask = (1u << kScaleBits) - 1;

// Per-table data laid out for the hot loops:
//   encoder: per (table, value) the ryg_rans "fast encode" constants (reciprocal multiply instead of
//            a division; bit-identical to ((x / f) << 16) + x % f + start for 31-bit states);
//   decoder: the cdf as 32 x uint16 per table (entries past the table padded) for a SIMD symbol search.
struct EncSym {
    uint32_t x_max, rcp_freq, bias;
    uint16_t cmpl_freq, rcp_shift;
};

struct CdfGroup {
    int n = 0, stride = 0;
    std::vector<int32_t> cdf, sizes, offsets;
    std::vector<EncSym> esym;          // [n][stride]
    std::vector<uint16_t> dcdf;        // [n][32]: cdf[1..], biased by 0x8000 for signed compares
    std::vector<uint32_t> dmask;       // [n]: valid-lane mask (entries 1..max_value)
};

void build_fast_tables(CdfGroup& g)
{
    g.esym.assign((size_t)g.n * g.stride, EncSym{0, 0, 0, 0, 0});
    g.dcdf.assign((size_t)g.n * 32, 0x7fff);
    g.dmask.assign(g.n, 0);
    for (int t = 0; t < g.n; ++t) {
        const int32_t* cdf = g.cdf.data() + (size_t)t * g.stride;
        const int nsym = g.sizes[t] - 1;                 // symbols 0..max_value (max_value = escape)
        for (int v = 0; v < nsym; ++v) {
            const uint32_t start = (uint32_t)cdf[v], freq = (uint32_t)(cdf[v + 1] - cdf[v]);
            EncSym& e = g.esym[(size_t)t * g.stride + v];
            e.x_max = ((1u << 23 >> 16) << 8) * freq;
            e.cmpl_freq = (uint16_t)((1u << 16) - freq);
            if (freq < 2) {
                e.rcp_freq = ~0u;
                e.rcp_shift = 0;
                e.bias = start + (1u << 16) - 1;
            } else {
                uint32_t shift = 0;
                while (freq > (1u << shift)) ++shift;
                e.rcp_freq = (uint32_t)(((1ull << (shift + 31)) + freq - 1) / freq);
                e.rcp_shift = (uint16_t)(shift - 1);
                e.bias = start;
            }
        }
        const int max_value = g.sizes[t] - 2;
        uint32_t mask = 0;
        for (int i = 1; i <= max_value && i <= 32; ++i) {
            g.dcdf[(size_t)t * 32 + (i - 1)] = (uint16_t)((uint32_t)cdf[i] ^ 0x8000u);
            mask |= 1u << (i - 1);
        }
        g.dmask[t] = mask;
    }
}

int add_group(std::vector<CdfGroup>& groups, const int32_t* cdf, int n, int stride, const int32_t* sizes,
              const int32_t* offsets)
{
    if (!cdf || !sizes || !offsets || n <= 0 || stride < 3 || stride > 34) {
        dcvc::set_error("add_cdf: bad table (n=%d stride=%d, stride must be 3..34)", n, stride);
        return dcvc::E_ARG;
    }
    CdfGroup g;
    g.n = n;
    g.stride = stride;
    g.cdf.assign(cdf, cdf + (size_t)n * stride);
    g.sizes.assign(sizes, sizes + n);
    g.offsets.assign(offsets, offsets + n);
    for (int i = 0; i < n; ++i) {
        if (g.sizes[i] < 3 || g.sizes[i] > stride) {
            dcvc::set_error("add_cdf: table %d has size %d (stride %d)", i, g.sizes[i], stride);
            return dcvc::E_ARG;
        }
        const int32_t* c = g.cdf.data() + (size_t)i * stride;
        bool ok = c[0] == 0 && c[g.sizes[i] - 1] == (1 << kScaleBits);
        for (int j = 0; j + 1 < g.sizes[i]; ++j) ok = ok && c[j + 1] > c[j];
        if (!ok) {
            dcvc::set_error("add_cdf: table %d is not a strictly increasing 16-bit cdf", i);
            return dcvc::E_ARG;
        }
    }
    build_fast_tables(g);
    groups.push_back(std::move(g));
    return (int)groups.size() - 1;
}


// ---- sentinel scans: kept = entries whose (low) byte != 0xFF ------------------------------------
// 32-bit kept masks per chunk of 32 entries let the hot loops visit only the coded symbols.
inline uint32_t kept_mask32_u8(const uint8_t* p, int64_t n_left)
{
#if defined(__AVX2__)
    if (n_left >= 32) {
        const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(p));
        return ~(uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, _mm256_set1_epi8((char)0xff)));
    }
#endif
    uint32_t m = 0;
    const int k = (int)(n_left < 32 ? n_left : 32);
    for (int i = 0; i < k; ++i) m |= (uint32_t)(p[i] != 0xff) << i;
    return m;
}

inline uint32_t kept_mask32_i16(const int16_t* p, int64_t n_left)
{
#if defined(__AVX2__)
    if (n_left >= 32) {
        const __m256i ff = _mm256_set1_epi16(0x00ff);
        const __m256i a = _mm256_and_si256(_mm256_loadu_si256(reinterpret_cast<const __m256i*>(p)), ff);
        const __m256i b = _mm256_and_si256(_mm256_loadu_si256(reinterpret_cast<const __m256i*>(p + 16)), ff);
        // packus interleaves 128-bit lanes: fix the order with a 64-bit permute
        const __m256i lo = _mm256_permute4x64_epi64(_mm256_packus_epi16(a, b), 0xd8);
        return ~(uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(lo, _mm256_set1_epi8((char)0xff)));
    }
#endif
    uint32_t m = 0;
    const int k = (int)(n_left < 32 ? n_left : 32);
    for (int i = 0; i < k; ++i) m |= (uint32_t)((p[i] & 0xff) != 0xff) << i;
    return m;
}

// number of kept entries, the largest kept table index, and (if n0 >= 0) the position just after the
// n0-th kept entry (= first position of the second coder)
template <typename E, uint32_t (*MASK)(const E*, int64_t)>
inline void scan_kept(const E* p, int64_t n, int64_t n0, int64_t& kept, int& max_idx, int64_t& split)
{
    kept = 0;
    max_idx = 0;
    split = -1;
    for (int64_t i = 0; i < n; i += 32) {
        const uint32_t m = MASK(p + i, n - i);
        const int c = __builtin_popcount(m);
        if (split < 0 && n0 >= 0 && kept + c >= n0) {
            int64_t need = n0 - kept;       // entries of this chunk that still belong to coder 0
            uint32_t mm = m;
            int pos = 0;
            while (need > 0) {
                pos = __builtin_ctz(mm) + 1;
                mm &= mm - 1;
                --need;
            }
            split = i + pos;
        }
        kept += c;
        for (uint32_t mm = m; mm; mm &= mm - 1) {
            const int v = (int)(p[i + __builtin_ctz(mm)] & 0xff);
            max_idx = v > max_idx ? v : max_idx;
        }
    }
    if (split < 0) split = n;
}

// one background thread executing jobs in order
class Worker {
public:
    Worker() : th_([this] { run(); }) {}
    ~Worker()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        th_.join();
    }
    void post(std::function<void()> f)
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            q_.push_back(std::move(f));
            ++pending_;
        }
        cv_.notify_all();
    }
    void wait_idle()
    {
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [this] { return pending_ == 0; });
    }

private:
    void run()
    {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [this] { return stop_ || !q_.empty(); });
                if (q_.empty()) return;
                f = std::move(q_.front());
                q_.pop_front();
            }
            f();
            {
                std::lock_guard<std::mutex> lk(m_);
                --pending_;
            }
            done_.notify_all();
        }
    }
    std::mutex m_;
    std::condition_variable cv_, done_;
    std::deque<std::function<void()>> q_;
    int pending_ = 0;
    bool stop_ = false;
    std::thread th_;
};

// ------------------------------------------------------------------ encoder half
struct EncTask {
    bool is_z = false;
    std::shared_ptr<std::vector<int16_t>> y;   // whole array incl. sentinels, shared by both halves
    int64_t begin = 0, end = 0;                // this half's range of y / z
    std::shared_ptr<std::vector<int8_t>> z;
    int group = 0, start_offset = 0, per_channel = 1;
    int64_t count = 0;                         // symbols actually coded (sentinels excluded)
};

inline void put_bits(uint32_t& r, uint8_t*& ptr, uint32_t val)
{
    constexpr uint32_t x_max = (1u << (kScaleBits - kBypassBits)) << kRenormShift;
    while (r >= x_max) {
        *(--ptr) = (uint8_t)(r & 0xff);
        r >>= 8;
    }
    r = (r << kBypassBits) | val;
}

inline void put_symbol(uint32_t& r, uint8_t*& ptr, const EncSym& e)
{
    uint32_t x = r;
    if (x >= e.x_max) {
        *(--ptr) = (uint8_t)(x & 0xff);
        x >>= 8;
        if (x >= e.x_max) {
            *(--ptr) = (uint8_t)(x & 0xff);
            x >>= 8;
        }
    }
    const uint32_t q = (uint32_t)(((uint64_t)x * e.rcp_freq) >> 32) >> e.rcp_shift;
    r = x + e.bias + q * e.cmpl_freq;
}

inline void encode_symbol(uint32_t& r, uint8_t*& ptr, int32_t symbol, const CdfGroup& g, int cdf_idx)
{
    const int32_t max_value = g.sizes[cdf_idx] - 2;
    int32_t value = symbol - g.offsets[cdf_idx];
    if ((uint32_t)value < (uint32_t)max_value) {         // common case: inside the table
        put_symbol(r, ptr, g.esym[(size_t)cdf_idx * g.stride + value]);
        return;
    }
    uint32_t raw;
    if (value < 0)
        raw = (uint32_t)(-2 * value - 1);
    else
        raw = (uint32_t)(2 * (value - max_value));
    uint8_t bins[48];
    int nb = 0;
    int n_bypass = 0;
    while ((raw >> (n_bypass * kBypassBits)) != 0) ++n_bypass;
    int v = n_bypass;
    while (v >= kBypassMax) {
        bins[nb++] = kBypassMax;
        v -= kBypassMax;
    }
    bins[nb++] = (uint8_t)v;
    for (int j = 0; j < n_bypass; ++j) bins[nb++] = (uint8_t)((raw >> (j * kBypassBits)) & kBypassMax);
    for (int i = nb - 1; i >= 0; --i) put_bits(r, ptr, bins[i]);
    put_symbol(r, ptr, g.esym[(size_t)cdf_idx * g.stride + max_value]);
}

struct EncHalf {
    std::vector<EncTask> tasks;
    std::vector<uint8_t> stream, scratch;
    void flush(const std::vector<CdfGroup>& groups)
    {
        size_t total = 0;
        for (auto& t : tasks) total += (size_t)t.count;
        stream.clear();
        if (total == 0) return;
        scratch.resize(total * 4 + 16);
        uint8_t* end = scratch.data() + scratch.size();
        uint8_t* ptr = end;
        uint32_t r = kRansL;
        for (auto it = tasks.rbegin(); it != tasks.rend(); ++it) {
            const CdfGroup& g = groups[it->group];
            if (it->is_z) {
                const int8_t* z = it->z->data();
                for (int64_t i = it->end - 1; i >= it->begin; --i)
                    encode_symbol(r, ptr, z[i], g, (int)((i - it->begin) / it->per_channel) + it->start_offset);
            } else {
                const int16_t* y = it->y->data();
                // walk [begin, end) backwards in 32-entry chunks aligned to `begin`, visiting kept entries only
                const int64_t len = it->end - it->begin;
                for (int64_t c0 = ((len - 1) / 32) * 32; c0 >= 0 && len > 0; c0 -= 32) {
                    const int16_t* q = y + it->begin + c0;
                    uint32_t m = kept_mask32_i16(q, len - c0);
                    while (m) {
                        const int b = 31 - __builtin_clz(m);
                        m &= ~(1u << b);
                        const int32_t cs = q[b];
                        encode_symbol(r, ptr, cs >> 8, g, cs & 0xff);
                    }
                }
            }
        }
        ptr -= 4;
        ptr[0] = (uint8_t)(r >> 0);
        ptr[1] = (uint8_t)(r >> 8);
        ptr[2] = (uint8_t)(r >> 16);
        ptr[3] = (uint8_t)(r >> 24);
        stream.assign(ptr, end);
    }
};

// ------------------------------------------------------------------ decoder half
struct DecHalf {
    std::vector<uint8_t> buf;
    size_t pos = 0;
    uint32_t state = 0;
    bool overrun = false;
    void init(const uint8_t* s, size_t n, bool reversed)
    {
        buf.assign(n + 16, 0);
        if (reversed)
            std::reverse_copy(s, s + n, buf.begin());
        else
            std::copy(s, s + n, buf.begin());
        state = (uint32_t)buf[0] | ((uint32_t)buf[1] << 8) | ((uint32_t)buf[2] << 16) | ((uint32_t)buf[3] << 24);
        pos = 4;
        overrun = n < 4;
    }
    inline uint8_t next()
    {
        if (pos >= buf.size()) {
            overrun = true;
            return 0;
        }
        return buf[pos++];
    }
    inline uint32_t get_bits()
    {
        const uint32_t val = state & ((1u << kBypassBits) - 1);
        state >>= kBypassBits;
        if (state < kRansL) state = (state << 8) | next();
        return val;
    }
    inline int8_t decode(const CdfGroup& g, int cdf_idx)
    {
        const int32_t* cdf = g.cdf.data() + (size_t)cdf_idx * g.stride;
        const int32_t max_value = g.sizes[cdf_idx] - 2;
        const uint32_t cum = state & kMask;
        int s;
#if defined(__AVX2__)
        {   // s = #{ i in 1..max_value : cdf[i] <= cum }  (branch-free; tables have at most 33 entries)
            const __m256i c = _mm256_set1_epi16((short)(cum ^ 0x8000u));
            const __m256i* t = reinterpret_cast<const __m256i*>(g.dcdf.data() + (size_t)cdf_idx * 32);
            const __m256i gt0 = _mm256_cmpgt_epi16(_mm256_loadu_si256(t), c);        // cdf[i] > cum
            const __m256i gt1 = _mm256_cmpgt_epi16(_mm256_loadu_si256(t + 1), c);
            const uint32_t m0 = (uint32_t)_mm256_movemask_epi8(_mm256_packs_epi16(gt0, gt1));
            // packs interleaves 128-bit lanes: bits [0..7]=gt0.lo, [8..15]=gt1.lo, [16..23]=gt0.hi, [24..31]=gt1.hi
            const uint32_t le = ~(((m0 & 0xffu)) | ((m0 >> 8) & 0xff00u) | ((m0 & 0xff00u) << 8) | (m0 & 0xff000000u));
            s = __builtin_popcount(le & g.dmask[cdf_idx]);
        }
#else
        s = 0;
        while (s < max_value && (uint32_t)cdf[s + 1] <= cum) ++s;
#endif
        const uint32_t start = (uint32_t)cdf[s], freq = (uint32_t)(cdf[s + 1] - cdf[s]);
        state = freq * (state >> kScaleBits) + cum - start;
        if (state < kRansL) {
            state = (state << 8) | next();
            if (state < kRansL) state = (state << 8) | next();
        }
        int32_t value = s;
        if (value == max_value) {
            int32_t val = (int32_t)get_bits();
            int32_t n_bypass = val;
            while (val == kBypassMax && !overrun) {
                val = (int32_t)get_bits();
                n_bypass += val;
            }
            int32_t raw = 0;
            for (int j = 0; j < n_bypass && j < 16; ++j) {
                val = (int32_t)get_bits();
                raw |= val << (j * kBypassBits);
            }
            value = raw >> 1;
            if (raw & 1)
                value = -value - 1;
            else
                value += max_value;
        }
        return (int8_t)(value + g.offsets[cdf_idx]);
    }
};

}  // namespace

struct dcvc_rans_enc {
    std::vector<CdfGroup> groups;
    bool two = false;
    EncHalf half[2];
    Worker worker[2];
    std::vector<uint8_t> merged;
    bool flushed = false;
};

struct dcvc_rans_dec {
    std::vector<CdfGroup> groups;
    bool two = false;
    DecHalf half[2];
    Worker worker[2];
    std::vector<int8_t> out;
};

extern "C" {

dcvc_rans_enc* dcvc_rans_enc_create(void) { return new (std::nothrow) dcvc_rans_enc(); }
void dcvc_rans_enc_destroy(dcvc_rans_enc* e) { delete e; }

int dcvc_rans_enc_add_cdf(dcvc_rans_enc* e, const int32_t* cdf, int n, int stride, const int32_t* sizes,
                          const int32_t* offsets)
{
    DCVC_REQUIRE(e, "dcvc_rans_enc_add_cdf: null coder");
    return add_group(e->groups, cdf, n, stride, sizes, offsets);
}

void dcvc_rans_enc_set_use_two(dcvc_rans_enc* e, int two)
{
    if (e) e->two = two != 0;
}

int dcvc_rans_enc_reset(dcvc_rans_enc* e)
{
    DCVC_REQUIRE(e, "dcvc_rans_enc_reset: null coder");
    e->worker[0].wait_idle();
    e->worker[1].wait_idle();
    e->half[0].tasks.clear();
    e->half[1].tasks.clear();
    e->half[0].stream.clear();
    e->half[1].stream.clear();
    e->merged.clear();
    e->flushed = false;
    return 0;
}

int dcvc_rans_enc_encode_y(dcvc_rans_enc* e, const int16_t* symbols, int64_t n, int group)
{
    DCVC_REQUIRE(e && (symbols || n == 0) && n >= 0, "dcvc_rans_enc_encode_y: bad arguments");
    DCVC_REQUIRE(group >= 0 && group < (int)e->groups.size(), "dcvc_rans_enc_encode_y: unknown cdf group %d", group);
    const CdfGroup& g = e->groups[group];
    auto buf = std::make_shared<std::vector<int16_t>>(symbols, symbols + n);   // inputs are copied on entry
    const int16_t* y = buf->data();
    int64_t kept = 0, split = n;   // split: first position that belongs to coder 1
    int max_idx = 0;
    scan_kept<int16_t, kept_mask32_i16>(y, n, -1, kept, max_idx, split);
    DCVC_REQUIRE(max_idx < g.n, "dcvc_rans_enc_encode_y: cdf index %d out of range (%d tables)", max_idx, g.n);
    const int64_t n0 = e->two ? kept / 2 : kept;
    if (e->two) {
        int64_t k2;
        int mx2;
        scan_kept<int16_t, kept_mask32_i16>(y, n, n0, k2, mx2, split);
        if (n0 == 0) split = 0;
    } else {
        split = n;
    }
    EncTask t0;
    t0.group = group;
    t0.y = buf;
    t0.begin = 0;
    t0.end = split;
    t0.count = n0;
    e->half[0].tasks.push_back(std::move(t0));
    if (e->two) {
        EncTask t1;
        t1.group = group;
        t1.y = buf;
        t1.begin = split;
        t1.end = n;
        t1.count = kept - n0;
        e->half[1].tasks.push_back(std::move(t1));
    }
    return 0;
}

int dcvc_rans_enc_encode_z(dcvc_rans_enc* e, const int8_t* symbols, int64_t n, int group, int start_offset,
                           int per_channel_size)
{
    DCVC_REQUIRE(e && (symbols || n == 0) && n >= 0 && per_channel_size > 0, "dcvc_rans_enc_encode_z: bad arguments");
    DCVC_REQUIRE(group >= 0 && group < (int)e->groups.size(), "dcvc_rans_enc_encode_z: unknown cdf group %d", group);
    const CdfGroup& g = e->groups[group];
    DCVC_REQUIRE(start_offset >= 0 && start_offset + (n + per_channel_size - 1) / per_channel_size <= g.n,
                 "dcvc_rans_enc_encode_z: channels exceed the cdf group");
    auto buf = std::make_shared<std::vector<int8_t>>(symbols, symbols + n);
    const int64_t n0 = e->two ? n / 2 : n;
    EncTask t0;
    t0.is_z = true;
    t0.group = group;
    t0.start_offset = start_offset;
    t0.per_channel = per_channel_size;
    t0.z = buf;
    t0.begin = 0;
    t0.end = n0;
    t0.count = n0;
    e->half[0].tasks.push_back(std::move(t0));
    if (e->two) {
        EncTask t1;
        t1.is_z = true;
        t1.group = group;
        t1.start_offset = start_offset + (int)(n0 / per_channel_size);
        t1.per_channel = per_channel_size;
        t1.z = buf;
        t1.begin = n0;
        t1.end = n;
        t1.count = n - n0;
        e->half[1].tasks.push_back(std::move(t1));
    }
    return 0;
}

int dcvc_rans_enc_flush(dcvc_rans_enc* e)
{
    DCVC_REQUIRE(e, "dcvc_rans_enc_flush: null coder");
    e->worker[0].post([e] { e->half[0].flush(e->groups); });
    if (e->two) e->worker[1].post([e] { e->half[1].flush(e->groups); });
    e->flushed = true;
    return 0;
}

int64_t dcvc_rans_enc_get_stream(dcvc_rans_enc* e, const uint8_t** data)
{
    if (!e || !data) {
        dcvc::set_error("dcvc_rans_enc_get_stream: null pointer");
        return dcvc::E_ARG;
    }
    if (!e->flushed) {
        dcvc::set_error("dcvc_rans_enc_get_stream: flush() has not been called");
        return dcvc::E_STREAM;
    }
    e->worker[0].wait_idle();
    e->worker[1].wait_idle();
    const std::vector<uint8_t>& s0 = e->half[0].stream;
    if (!e->two) {
        *data = s0.data();
        return (int64_t)s0.size();
    }
    const std::vector<uint8_t>& s1 = e->half[1].stream;
    const size_t n0 = s0.size(), n1 = s1.size();
    size_t identical = 0;
    const size_t check = std::min<size_t>(std::min(n0, n1), 8);
    for (size_t i = 0; i < check; ++i) {
        if (s0[n0 - 1 - i] != 0 || s1[n1 - 1 - i] != 0) break;
        ++identical;
    }
    if (identical == 0 && n0 > 0 && n1 > 0 && s0[n0 - 1] == s1[n1 - 1]) identical = 1;
    e->merged.resize(n0 + n1 - identical);
    std::copy(s0.begin(), s0.end(), e->merged.begin());
    std::reverse_copy(s1.begin(), s1.end() - identical, e->merged.begin() + n0);
    *data = e->merged.data();
    return (int64_t)e->merged.size();
}

dcvc_rans_dec* dcvc_rans_dec_create(void) { return new (std::nothrow) dcvc_rans_dec(); }
void dcvc_rans_dec_destroy(dcvc_rans_dec* d) { delete d; }

int dcvc_rans_dec_add_cdf(dcvc_rans_dec* d, const int32_t* cdf, int n, int stride, const int32_t* sizes,
                          const int32_t* offsets)
{
    DCVC_REQUIRE(d, "dcvc_rans_dec_add_cdf: null coder");
    return add_group(d->groups, cdf, n, stride, sizes, offsets);
}

void dcvc_rans_dec_set_use_two(dcvc_rans_dec* d, int two)
{
    if (d) d->two = two != 0;
}

int dcvc_rans_dec_set_stream(dcvc_rans_dec* d, const uint8_t* data, int64_t n)
{
    DCVC_REQUIRE(d && data && n >= 4, "dcvc_rans_dec_set_stream: stream of %lld bytes is too short", (long long)n);
    d->worker[0].wait_idle();
    d->worker[1].wait_idle();
    d->half[0].init(data, (size_t)n, false);
    if (d->two) d->half[1].init(data, (size_t)n, true);
    return 0;
}

int dcvc_rans_dec_decode_y(dcvc_rans_dec* d, const uint8_t* indexes, int64_t n, int group)
{
    DCVC_REQUIRE(d && (indexes || n == 0) && n >= 0, "dcvc_rans_dec_decode_y: bad arguments");
    DCVC_REQUIRE(group >= 0 && group < (int)d->groups.size(), "dcvc_rans_dec_decode_y: unknown cdf group %d", group);
    d->worker[0].wait_idle();
    d->worker[1].wait_idle();
    auto idx = std::make_shared<std::vector<uint8_t>>(indexes, indexes + n);
    const CdfGroup* g = &d->groups[group];
    const uint8_t* ip = idx->data();
    int64_t kept = 0, split = n;   // split: first position handled by coder 1
    int max_idx = 0;
    scan_kept<uint8_t, kept_mask32_u8>(ip, n, -1, kept, max_idx, split);
    DCVC_REQUIRE(max_idx < g->n, "dcvc_rans_dec_decode_y: cdf index %d out of range (%d tables)", max_idx, g->n);
    d->out.assign((size_t)n, 0);
    if (d->two) {
        int64_t k2;
        int mx2;
        scan_kept<uint8_t, kept_mask32_u8>(ip, n, kept / 2, k2, mx2, split);
        if (kept / 2 == 0) split = 0;
    } else {
        split = n;
    }
    int8_t* out = d->out.data();
    auto job = [idx, g, out](DecHalf* h, int64_t a, int64_t b) {
        const uint8_t* q = idx->data();
        DecHalf local = std::move(*h);          // keep the coder state on this thread's stack
        for (int64_t c0 = a; c0 < b; c0 += 32) {   // visit kept entries only
            uint32_t m = kept_mask32_u8(q + c0, b - c0);
            while (m) {
                const int bit = __builtin_ctz(m);
                m &= m - 1;
                out[c0 + bit] = local.decode(*g, q[c0 + bit]);
            }
        }
        *h = std::move(local);
    };
    d->worker[0].post([=] { job(&d->half[0], 0, split); });
    if (d->two) d->worker[1].post([=] { job(&d->half[1], split, n); });
    return 0;
}

int dcvc_rans_dec_decode_z(dcvc_rans_dec* d, int64_t total, int group, int start_offset, int per_channel_size)
{
    DCVC_REQUIRE(d && total >= 0 && per_channel_size > 0, "dcvc_rans_dec_decode_z: bad arguments");
    DCVC_REQUIRE(group >= 0 && group < (int)d->groups.size(), "dcvc_rans_dec_decode_z: unknown cdf group %d", group);
    const CdfGroup* g = &d->groups[group];
    DCVC_REQUIRE(start_offset >= 0 && start_offset + (total + per_channel_size - 1) / per_channel_size <= g->n,
                 "dcvc_rans_dec_decode_z: channels exceed the cdf group");
    d->worker[0].wait_idle();
    d->worker[1].wait_idle();
    d->out.assign((size_t)total, 0);
    int8_t* out = d->out.data();
    const int64_t n0 = d->two ? total / 2 : total;
    auto job = [g, out, per_channel_size](DecHalf* h, int64_t base, int64_t cnt, int start) {
        for (int64_t i = 0; i < cnt; ++i) out[base + i] = h->decode(*g, (int)(i / per_channel_size) + start);
    };
    d->worker[0].post([=] { job(&d->half[0], 0, n0, start_offset); });
    if (d->two) {
        const int start1 = start_offset + (int)(n0 / per_channel_size);
        d->worker[1].post([=] { job(&d->half[1], n0, total - n0, start1); });
    }
    return 0;
}

int64_t dcvc_rans_dec_get(dcvc_rans_dec* d, int8_t* out, int64_t capacity)
{
    if (!d || !out) {
        dcvc::set_error("dcvc_rans_dec_get: null pointer");
        return dcvc::E_ARG;
    }
    d->worker[0].wait_idle();
    d->worker[1].wait_idle();
    if (d->half[0].overrun || (d->two && d->half[1].overrun)) {
        dcvc::set_error("dcvc_rans_dec_get: bit stream exhausted (corrupt or truncated stream)");
        return dcvc::E_STREAM;
    }
    if ((int64_t)d->out.size() > capacity) {
        dcvc::set_error("dcvc_rans_dec_get: output buffer too small (%lld > %lld)", (long long)d->out.size(),
                        (long long)capacity);
        return dcvc::E_ARG;
    }
    std::memcpy(out, d->out.data(), d->out.size());
    return (int64_t)d->out.size();
}

int dcvc_pmf_to_quantized_cdf(const float* pmf, int n, int precision, uint32_t* out)
{
    DCVC_REQUIRE(pmf && out && n > 0 && precision > 0 && precision <= 16, "dcvc_pmf_to_quantized_cdf: bad arguments");
    out[0] = 0;
    for (int i = 0; i < n; ++i) out[i + 1] = (uint32_t)(std::round(pmf[i] * (float)(1 << precision)) + 0.5);
    uint32_t total = 0;
    for (int i = 0; i <= n; ++i) total += out[i];
    DCVC_REQUIRE(total > 0, "dcvc_pmf_to_quantized_cdf: empty pmf");
    for (int i = 0; i <= n; ++i) out[i] = (uint32_t)((((uint64_t)1 << precision) * out[i]) / total);
    for (int i = 1; i <= n; ++i) out[i] += out[i - 1];
    out[n] = 1u << precision;
    for (int i = 0; i < n; ++i) {
        if (out[i] != out[i + 1]) continue;
        uint32_t best_freq = ~0u;
        int best = -1;
        for (int j = 0; j < n; ++j) {
            const uint32_t f = out[j + 1] - out[j];
            if (f > 1 && f < best_freq) {
                best_freq = f;
                best = j;
            }
        }
        DCVC_REQUIRE(best >= 0, "dcvc_pmf_to_quantized_cdf: cannot make every symbol codable");
        if (best < i) {
            for (int j = best + 1; j <= i; ++j) out[j]--;
        } else {
            for (int j = i + 1; j <= best; ++j) out[j]++;
        }
    }
    return 0;
}

}  // extern "C"
