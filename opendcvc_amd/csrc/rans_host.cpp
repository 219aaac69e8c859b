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
//   decoder: per table one cache-aligned record: (start | freq << 16) per value and a 128-entry
//            first-guess table indexed by the top bits of the cumulative slot.  The guess is the value
//            that owns the lower edge of the slot's bucket; a (well predicted, mostly zero-trip) loop
//            walks up from there.  The guess table holds the guessed value's (start | freq << 16) itself
//            next to its index, so the state -> state dependency chain carries ONE L1 load (the index
//            is loaded beside it, off the chain); against two loads: -14 % decode time on real 1080p
//            symbols, and ~17 cycles for a SIMD compare / movemask / popcount search.
struct EncSym {
    uint32_t x_max, rcp_freq, bias;
    uint16_t cmpl_freq, rcp_shift;
};

constexpr int kLutBits = 7;
struct alignas(64) DecTable {
    uint32_t sym[36];                  // values 0..max_value (max_value = escape); rest unused
    uint32_t lute[1 << kLutBits];      // the guessed value's (start | freq << 16) itself: ONE load on the state chain
    uint8_t lut[1 << kLutBits];
    int16_t offset;
    uint16_t max_value;
};

struct CdfGroup {
    int n = 0, stride = 0;
    std::vector<int32_t> cdf, sizes, offsets;
    std::vector<EncSym> esym;          // [n][stride]
    std::vector<DecTable> dtab;        // [n]
};

void build_fast_tables(CdfGroup& g)
{
    g.esym.assign((size_t)g.n * g.stride, EncSym{0, 0, 0, 0, 0});
    g.dtab.assign((size_t)g.n, DecTable{});
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
        DecTable& d = g.dtab[t];
        d.offset = (int16_t)g.offsets[t];
        d.max_value = (uint16_t)(nsym - 1);
        for (int v = 0; v < nsym; ++v) d.sym[v] = (uint32_t)cdf[v] | ((uint32_t)(cdf[v + 1] - cdf[v]) << 16);
        int v = 0;
        for (int b = 0; b < (1 << kLutBits); ++b) {
            const int32_t edge = b << (kScaleBits - kLutBits);
            while (cdf[v + 1] <= edge) ++v;              // cdf[nsym] == 65536 > every edge
            d.lut[b] = (uint8_t)v;
            d.lute[b] = d.sym[v];
        }
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

// One pass over an index / packed-symbol array: kept entries per 32-entry chunk, their total and the
// largest kept table index.  split_after(k) = the position just after the k-th kept entry (the first
// position that belongs to the second coder when the first one takes k symbols).
struct KeptScan {
    std::vector<uint8_t> per_chunk;
    int64_t kept = 0, n = 0;
    int max_idx = 0;
};

inline void scan_kept_u8(const uint8_t* p, int64_t n, KeptScan& k)
{
    k.per_chunk.resize((size_t)((n + 31) / 32));
    k.kept = 0;
    k.n = n;
    int64_t i = 0;
    int mx = 0;
#if defined(__AVX2__)
    const __m256i ff = _mm256_set1_epi8((char)0xff);
    __m256i vmax = _mm256_setzero_si256();
    for (; i + 32 <= n; i += 32) {
        const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(p + i));
        const __m256i eq = _mm256_cmpeq_epi8(v, ff);
        vmax = _mm256_max_epu8(vmax, _mm256_andnot_si256(eq, v));
        const int c = __builtin_popcount(~(uint32_t)_mm256_movemask_epi8(eq));
        k.per_chunk[(size_t)(i >> 5)] = (uint8_t)c;
        k.kept += c;
    }
    alignas(32) uint8_t lanes[32];
    _mm256_store_si256(reinterpret_cast<__m256i*>(lanes), vmax);
    for (int j = 0; j < 32; ++j) mx = lanes[j] > mx ? lanes[j] : mx;
#endif
    for (; i < n; i += 32) {
        int c = 0;
        for (int64_t j = i; j < n && j < i + 32; ++j)
            if (p[j] != 0xff) {
                ++c;
                mx = p[j] > mx ? p[j] : mx;
            }
        k.per_chunk[(size_t)(i >> 5)] = (uint8_t)c;
        k.kept += c;
    }
    k.max_idx = mx;
}

inline void scan_kept_i16(const int16_t* p, int64_t n, KeptScan& k)
{
    k.per_chunk.resize((size_t)((n + 31) / 32));
    k.kept = 0;
    k.n = n;
    int64_t i = 0;
    int mx = 0;
#if defined(__AVX2__)
    const __m256i lo = _mm256_set1_epi16(0x00ff);
    __m256i vmax = _mm256_setzero_si256();
    for (; i + 32 <= n; i += 32) {
        const __m256i a = _mm256_and_si256(_mm256_loadu_si256(reinterpret_cast<const __m256i*>(p + i)), lo);
        const __m256i b = _mm256_and_si256(_mm256_loadu_si256(reinterpret_cast<const __m256i*>(p + i + 16)), lo);
        const __m256i ea = _mm256_cmpeq_epi16(a, lo), eb = _mm256_cmpeq_epi16(b, lo);
        vmax = _mm256_max_epi16(vmax, _mm256_max_epi16(_mm256_andnot_si256(ea, a), _mm256_andnot_si256(eb, b)));
        const uint32_t skipped = (uint32_t)_mm256_movemask_epi8(_mm256_packs_epi16(ea, eb));   // lane order irrelevant
        const int c = 32 - __builtin_popcount(skipped);
        k.per_chunk[(size_t)(i >> 5)] = (uint8_t)c;
        k.kept += c;
    }
    alignas(32) int16_t lanes[16];
    _mm256_store_si256(reinterpret_cast<__m256i*>(lanes), vmax);
    for (int j = 0; j < 16; ++j) mx = lanes[j] > mx ? lanes[j] : mx;
#endif
    for (; i < n; i += 32) {
        int c = 0;
        for (int64_t j = i; j < n && j < i + 32; ++j) {
            const int v = p[j] & 0xff;
            if (v != 0xff) {
                ++c;
                mx = v > mx ? v : mx;
            }
        }
        k.per_chunk[(size_t)(i >> 5)] = (uint8_t)c;
        k.kept += c;
    }
    k.max_idx = mx;
}

template <typename E, uint32_t (*MASK)(const E*, int64_t)>
inline int64_t split_after(const KeptScan& k, const E* p, int64_t take)
{
    if (take <= 0) return 0;
    if (take >= k.kept) return k.n;
    int64_t seen = 0;
    size_t c = 0;
    while (seen + k.per_chunk[c] < take) seen += k.per_chunk[c++];
    uint32_t m = MASK(p + (int64_t)c * 32, k.n - (int64_t)c * 32);
    int pos = 0;
    for (int64_t need = take - seen; need > 0; --need) {
        pos = __builtin_ctz(m) + 1;
        m &= m - 1;
    }
    return (int64_t)c * 32 + pos;
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
    const int16_t* y = nullptr;                // whole array incl. sentinels, shared by both halves
    std::shared_ptr<std::vector<int16_t>> y_own;   // set when the array was copied on entry
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
    // first renormalisation byte without a branch (taken for roughly every fourth symbol, i.e. badly
    // predicted): always store below the write pointer, move the pointer only if the byte was needed
    ptr[-1] = (uint8_t)(x & 0xff);
    const bool spill = x >= e.x_max;
    ptr -= spill;
    x = spill ? x >> 8 : x;
    if (__builtin_expect(x >= e.x_max, 0)) {
        *(--ptr) = (uint8_t)(x & 0xff);
        x >>= 8;
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
                const int8_t* z = it->z->data() + it->begin;
                const int64_t len = it->end - it->begin;
                int64_t i = len - 1;
                for (int64_t ch = i / it->per_channel; i >= 0; --ch)       // channel by channel, backwards
                    for (const int64_t first = ch * it->per_channel; i >= first; --i)
                        encode_symbol(r, ptr, z[i], g, (int)ch + it->start_offset);
            } else {
                const int16_t* y = it->y;
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
// The coder state while a run of symbols is decoded: lives on the decoding thread's stack (so it
// stays in registers; the int8 output stores could alias anything reachable through a pointer).
struct DecCursor {
    uint32_t x;
    const uint8_t* cur;
    const uint8_t* end;
    bool overrun;

    inline uint32_t next()
    {
        if (cur >= end) {
            overrun = true;
            return 0;
        }
        return *cur++;
    }
    inline uint32_t get_bits()
    {
        const uint32_t val = x & ((1u << kBypassBits) - 1);
        x >>= kBypassBits;
        if (x < kRansL) x = (x << 8) | next();
        return val;
    }
    inline int32_t decode(const DecTable& t)
    {
        const uint32_t cum = x & kMask;
        uint32_t s = t.lut[cum >> (kScaleBits - kLutBits)];
        uint32_t e = t.lute[cum >> (kScaleBits - kLutBits)];
        uint32_t d = cum - (e & kMask);
        while (d >= (e >> 16)) {              // cum lies past this value's range: walk up
            e = t.sym[++s];
            d = cum - (e & kMask);
        }
        x = (e >> 16) * (x >> kScaleBits) + d;
        if (x < kRansL) {
            x = (x << 8) | next();
            if (x < kRansL) x = (x << 8) | next();
        }
        int32_t value = (int32_t)s;
        if (s == t.max_value) {               // escape: Exp-Golomb-like bypass bits follow
            // (a valid stream carries at most 16 groups - 32 bits - per escape; a corrupt one may claim any number: the
            // count saturates, the value is assembled without signed overflow, and the decoder keeps its position rules)
            uint32_t val = get_bits();
            uint32_t n_bypass = val;
            while (val == (uint32_t)kBypassMax && !overrun) {
                val = get_bits();
                n_bypass = n_bypass < 1024u ? n_bypass + val : n_bypass;
            }
            uint32_t raw = 0;
            for (uint32_t j = 0; j < n_bypass && j < 16; ++j) raw |= get_bits() << (j * kBypassBits);
            value = (int32_t)(raw >> 1);
            if (raw & 1)
                value = -value - 1;
            else
                value += t.max_value;
        }
        return value + t.offset;
    }
};

struct DecHalf {
    std::vector<uint8_t> buf;
    size_t pos = 0;
    uint32_t state = 0;
    bool overrun = false;
    int64_t decoded = 0;               // symbols decoded since init()
    void init(const uint8_t* s, size_t n, bool reversed)
    {
        decoded = 0;
        buf.assign(n + 16, 0);
        if (reversed)
            std::reverse_copy(s, s + n, buf.begin());
        else
            std::copy(s, s + n, buf.begin());
        state = (uint32_t)buf[0] | ((uint32_t)buf[1] << 8) | ((uint32_t)buf[2] << 16) | ((uint32_t)buf[3] << 24);
        pos = 4;
        overrun = n < 4;
    }
    DecCursor open() const { return DecCursor{state, buf.data() + pos, buf.data() + buf.size(), overrun}; }
    void close(const DecCursor& c)
    {
        state = c.x;
        pos = (size_t)(c.cur - buf.data());
        overrun = c.overrun;
    }

    // y symbols of positions [a, b): zero-fill, then decode the kept entries (index != 0xFF) in order
    void decode_range(const CdfGroup& g, const uint8_t* idx, int8_t* out, int64_t a, int64_t b)
    {
        if (b <= a) return;
        std::memset(out + a, 0, (size_t)(b - a));
        const DecTable* tab = g.dtab.data();
        DecCursor c = open();
        for (int64_t c0 = a; c0 < b; c0 += 32) {
            uint32_t m = kept_mask32_u8(idx + c0, b - c0);
            decoded += __builtin_popcount(m);
            while (m) {
                const int bit = __builtin_ctz(m);
                m &= m - 1;
                out[c0 + bit] = (int8_t)c.decode(tab[idx[c0 + bit]]);
            }
        }
        close(c);
    }

    // y symbols whose table indexes arrive already compacted (dcvc_prior_dec_index_compact): entries [a, b) of `idx`, one
    // symbol each, no sentinels to skip and nothing to zero-fill
    void decode_packed(const CdfGroup& g, const uint8_t* idx, int8_t* out, int64_t a, int64_t b)
    {
        if (b <= a) return;
        const DecTable* tab = g.dtab.data();
        DecCursor c = open();
        for (int64_t i = a; i < b; ++i) out[i] = (int8_t)c.decode(tab[idx[i]]);
        decoded += b - a;
        close(c);
    }

    // z symbols: `cnt` values, `per_channel` consecutive ones share table start, start + 1, ...
    void decode_channels(const CdfGroup& g, int8_t* out, int64_t cnt, int start, int per_channel)
    {
        const DecTable* tab = g.dtab.data() + start;
        DecCursor c = open();
        for (int64_t i = 0; i < cnt; ++tab) {
            const int64_t stop = std::min<int64_t>(cnt, i + per_channel);
            for (; i < stop; ++i) out[i] = (int8_t)c.decode(*tab);
        }
        decoded += cnt;
        close(c);
    }
};

}  // namespace

struct dcvc_rans_enc {
    std::vector<CdfGroup> groups;
    bool two = false;
    EncHalf half[2];
    Worker worker[2];
    std::vector<uint8_t> merged;
    KeptScan scan;
    bool flushed = false;
};

struct dcvc_rans_dec {
    std::vector<CdfGroup> groups;
    bool two = false;
    DecHalf half[2];
    Worker worker[2];
    std::vector<int8_t> out;
    std::vector<uint8_t> idx;      // copy of the indexes for the asynchronous decode_y
    KeptScan scan;
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

int dcvc_rans_enc_empty_cdf(dcvc_rans_enc* e)
{
    DCVC_REQUIRE(e, "dcvc_rans_enc_empty_cdf: null coder");
    e->worker[0].wait_idle();
    e->worker[1].wait_idle();
    e->groups.clear();
    return 0;
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

static int enc_add_y(dcvc_rans_enc* e, const int16_t* symbols, int64_t n, int group, bool copy, const char* who)
{
    DCVC_REQUIRE(e && (symbols || n == 0) && n >= 0, "%s: bad arguments", who);
    DCVC_REQUIRE(group >= 0 && group < (int)e->groups.size(), "%s: unknown cdf group %d", who, group);
    const CdfGroup& g = e->groups[group];
    std::shared_ptr<std::vector<int16_t>> own;
    const int16_t* y = symbols;
    if (copy) {
        own = std::make_shared<std::vector<int16_t>>(symbols, symbols + n);
        y = own->data();
    }
    scan_kept_i16(y, n, e->scan);
    DCVC_REQUIRE(e->scan.max_idx < g.n, "%s: cdf index %d out of range (%d tables)", who, e->scan.max_idx, g.n);
    const int64_t kept = e->scan.kept;
    const int64_t n0 = e->two ? kept / 2 : kept;
    const int64_t split = e->two ? split_after<int16_t, kept_mask32_i16>(e->scan, y, n0) : n;
    EncTask t0;
    t0.group = group;
    t0.y = y;
    t0.y_own = own;
    t0.begin = 0;
    t0.end = split;
    t0.count = n0;
    e->half[0].tasks.push_back(std::move(t0));
    if (e->two) {
        EncTask t1;
        t1.group = group;
        t1.y = y;
        t1.y_own = own;
        t1.begin = split;
        t1.end = n;
        t1.count = kept - n0;
        e->half[1].tasks.push_back(std::move(t1));
    }
    return 0;
}

int dcvc_rans_enc_encode_y(dcvc_rans_enc* e, const int16_t* symbols, int64_t n, int group)
{
    return enc_add_y(e, symbols, n, group, true, "dcvc_rans_enc_encode_y");
}

int dcvc_rans_enc_encode_y_borrowed(dcvc_rans_enc* e, const int16_t* symbols, int64_t n, int group)
{
    return enc_add_y(e, symbols, n, group, false, "dcvc_rans_enc_encode_y_borrowed");
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

int dcvc_rans_dec_empty_cdf(dcvc_rans_dec* d)
{
    DCVC_REQUIRE(d, "dcvc_rans_dec_empty_cdf: null coder");
    d->worker[0].wait_idle();
    d->worker[1].wait_idle();
    d->groups.clear();
    return 0;
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

// common part of the two y entry points: scan, split, run coder 1 on its worker and coder 0 either
// on its worker (asynchronous form) or on the calling thread (synchronous form)
static int dec_run_y(dcvc_rans_dec* d, const uint8_t* ip, int64_t n, int group, int8_t* out, bool inline0,
                     const char* who)
{
    DCVC_REQUIRE(group >= 0 && group < (int)d->groups.size(), "%s: unknown cdf group %d", who, group);
    const CdfGroup* g = &d->groups[group];
    scan_kept_u8(ip, n, d->scan);
    DCVC_REQUIRE(d->scan.max_idx < g->n, "%s: cdf index %d out of range (%d tables)", who, d->scan.max_idx, g->n);
    const int64_t split = d->two ? split_after<uint8_t, kept_mask32_u8>(d->scan, ip, d->scan.kept / 2) : n;
    if (d->two) d->worker[1].post([=] { d->half[1].decode_range(*g, ip, out, split, n); });
    if (inline0)
        d->half[0].decode_range(*g, ip, out, 0, split);
    else
        d->worker[0].post([=] { d->half[0].decode_range(*g, ip, out, 0, split); });
    return 0;
}

int dcvc_rans_dec_decode_y(dcvc_rans_dec* d, const uint8_t* indexes, int64_t n, int group)
{
    DCVC_REQUIRE(d && (indexes || n == 0) && n >= 0, "dcvc_rans_dec_decode_y: bad arguments");
    d->worker[0].wait_idle();
    d->worker[1].wait_idle();
    d->idx.assign(indexes, indexes + n);          // inputs are copied on entry
    d->out.resize((size_t)n);
    return dec_run_y(d, d->idx.data(), n, group, d->out.data(), false, "dcvc_rans_dec_decode_y");
}

int dcvc_rans_dec_decode_and_get_y(dcvc_rans_dec* d, const uint8_t* indexes, int64_t n, int group, int8_t* out)
{
    DCVC_REQUIRE(d && ((indexes && out) || n == 0) && n >= 0, "dcvc_rans_dec_decode_and_get_y: bad arguments");
    d->worker[0].wait_idle();
    d->worker[1].wait_idle();
    d->out.clear();
    const int rc = dec_run_y(d, indexes, n, group, out, true, "dcvc_rans_dec_decode_and_get_y");
    if (rc) return rc;
    d->worker[1].wait_idle();
    if (d->half[0].overrun || (d->two && d->half[1].overrun)) {
        dcvc::set_error("dcvc_rans_dec_decode_and_get_y: bit stream exhausted (corrupt or truncated stream)");
        return dcvc::E_STREAM;
    }
    return 0;
}

int dcvc_rans_dec_decode_compact(dcvc_rans_dec* d, const uint8_t* indexes, int64_t count, int group, int8_t* out)
{
    DCVC_REQUIRE(d && ((indexes && out) || count == 0) && count >= 0, "dcvc_rans_dec_decode_compact: bad arguments");
    DCVC_REQUIRE(group >= 0 && group < (int)d->groups.size(), "dcvc_rans_dec_decode_compact: unknown cdf group %d", group);
    d->worker[0].wait_idle();
    d->worker[1].wait_idle();
    d->out.clear();
    const CdfGroup* g = &d->groups[group];
    int mx = 0;
    for (int64_t i = 0; i < count; ++i) mx = indexes[i] > mx ? indexes[i] : mx;
    DCVC_REQUIRE(mx < g->n, "dcvc_rans_dec_decode_compact: cdf index %d out of range (%d tables)", mx, g->n);
    // the same split as dec_run_y: the first coder takes floor(count / 2) symbols, the second the rest
    const int64_t k0 = d->two ? count / 2 : count;
    if (d->two) d->worker[1].post([=] { d->half[1].decode_packed(*g, indexes, out, k0, count); });
    d->half[0].decode_packed(*g, indexes, out, 0, k0);
    d->worker[1].wait_idle();
    if (d->half[0].overrun || (d->two && d->half[1].overrun)) {
        dcvc::set_error("dcvc_rans_dec_decode_compact: bit stream exhausted (corrupt or truncated stream)");
        return dcvc::E_STREAM;
    }
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
    d->worker[0].post([=] { d->half[0].decode_channels(*g, out, n0, start_offset, per_channel_size); });
    if (d->two) {
        const int start1 = start_offset + (int)(n0 / per_channel_size);
        d->worker[1].post([=] { d->half[1].decode_channels(*g, out + n0, total - n0, start1, per_channel_size); });
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

int dcvc_rans_dec_check_end(dcvc_rans_dec* d)
{
    DCVC_REQUIRE(d, "dcvc_rans_dec_check_end: null coder");
    d->worker[0].wait_idle();
    d->worker[1].wait_idle();
    const int halves = d->two ? 2 : 1;
    int64_t consumed = 0, used = 0;
    for (int h = 0; h < halves; ++h) {
        const DecHalf& c = d->half[h];
        if (c.buf.size() < 16) continue;                     // set_stream() not called
        if (c.overrun) {
            dcvc::set_error("dcvc_rans_dec_check_end: bit stream exhausted (corrupt or truncated stream)");
            return dcvc::E_STREAM;
        }
        if (c.decoded == 0) continue;                        // an empty half has no stream of its own
        // decoding undoes the encoder step by step: after the last symbol the state is the encoder's initial one
        if (c.state != kRansL) {
            dcvc::set_error("dcvc_rans_dec_check_end: coder %d does not end in its initial state (corrupt or truncated "
                            "stream, or not every symbol of the frame has been decoded)", h);
            return dcvc::E_STREAM;
        }
        consumed += (int64_t)c.pos;
        ++used;
    }
    if (used == halves) {
        // every byte belongs to a coder: the two halves share at most 8 bytes (merge rule of get_encoded_stream)
        const int64_t n = (int64_t)d->half[0].buf.size() - 16;
        if (consumed < n || consumed > n + (d->two ? 8 : 0)) {
            dcvc::set_error("dcvc_rans_dec_check_end: %lld of %lld stream bytes consumed (trailing or missing bytes)",
                            (long long)consumed, (long long)n);
            return dcvc::E_STREAM;
        }
    }
    return 0;
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
