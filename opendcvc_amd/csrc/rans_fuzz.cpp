// rans_fuzz.cpp - robustness driver of the host rANS coder (rans_host.cpp), built with AddressSanitizer +
// UndefinedBehaviorSanitizer by `make asan` and run by tests/test_rans_robust.py on the CPU.
//
// The reference decodes a corrupt or truncated payload into garbage without any check
// (src/cpp/py_rans/rans.cpp:356-429 reads past the vector's end through a raw pointer); the drop-in must not read out of
// bounds and must say so: every decode call returns 0 or -4 (E_STREAM), dcvc_rans_dec_check_end() fails on every damaged
// stream, and the sanitizers see no invalid access / undefined behaviour on the way.
//
//   rans_fuzz_asan [rounds]      exit code 0 = all properties held
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "dcvc_amd.h"

namespace {

struct Rng {      // xorshift64*: deterministic, no library state
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 1) {}
    uint64_t next()
    {
        s ^= s >> 12;
        s ^= s << 25;
        s ^= s >> 27;
        return s * 0x2545F4914F6CDD1Dull;
    }
    int below(int n) { return (int)(next() % (uint64_t)n); }
};

int g_fail = 0;
static int g_compact_turn = 0;       // every other frame decodes its second half in the compacted form
#define EXPECT(cond, ...)                     \
    do {                                      \
        if (!(cond)) {                        \
            std::fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); \
            std::fprintf(stderr, __VA_ARGS__); \
            std::fprintf(stderr, "\n");       \
            ++g_fail;                         \
        }                                     \
    } while (0)

// n strictly increasing 16-bit cdfs of random sizes (3..stride entries, the last value of each = escape)
struct Tables {
    int n, stride;
    std::vector<int32_t> cdf, sizes, offsets;
};

Tables make_tables(Rng& r, int n, int stride)
{
    Tables t{n, stride, std::vector<int32_t>((size_t)n * stride, 0), std::vector<int32_t>(n), std::vector<int32_t>(n)};
    for (int i = 0; i < n; ++i) {
        const int size = 3 + r.below(stride - 2);           // entries incl. the leading 0 and the final 65536
        const int nsym = size - 1;
        std::vector<int32_t> f(nsym, 1);
        int left = 65536 - nsym;
        for (int k = 0; k < nsym; ++k) {                    // random split of the mass, peaked in the middle
            const int take = k + 1 == nsym ? left : r.below(left / (1 + (k != nsym / 2)) + 1);
            f[k] += take;
            left -= take;
        }
        int32_t* c = t.cdf.data() + (size_t)i * stride;
        c[0] = 0;
        for (int k = 0; k < nsym; ++k) c[k + 1] = c[k] + f[k];
        t.sizes[i] = size;
        t.offsets[i] = -(nsym / 2);
    }
    return t;
}

struct Frame {
    std::vector<int8_t> z;
    std::vector<int16_t> y[2];          // packed (sym << 8) | idx, idx 0xFF = skipped
    int per_channel;
};

Frame make_frame(Rng& r, const Tables& zt, const Tables& yt, int zn, int yn)
{
    Frame f;
    f.per_channel = (zn + zt.n - 1) / zt.n + r.below(zn);     // never more channels than the group has tables
    f.z.resize(zn);
    for (int i = 0; i < zn; ++i) f.z[i] = (int8_t)(r.below(41) - 20);          // beyond most tables: escapes
    for (int h = 0; h < 2; ++h) {
        f.y[h].resize(yn);
        for (int i = 0; i < yn; ++i) {
            const int idx = r.below(4) == 0 ? 0xFF : r.below(yt.n);
            const int sym = r.below(8) == 0 ? r.below(255) - 127 : r.below(9) - 4;
            f.y[h][i] = (int16_t)(sym * 256 + idx);
        }
    }
    return f;
}

std::vector<uint8_t> encode(dcvc_rans_enc* e, const Frame& f, int zg, int yg, int z_start)
{
    EXPECT(dcvc_rans_enc_reset(e) == 0, "reset");
    EXPECT(dcvc_rans_enc_encode_z(e, f.z.data(), (int64_t)f.z.size(), zg, z_start, f.per_channel) == 0, "encode_z: %s",
           dcvc_last_error());
    for (int h = 0; h < 2; ++h)
        EXPECT(dcvc_rans_enc_encode_y(e, f.y[h].data(), (int64_t)f.y[h].size(), yg) == 0, "encode_y: %s", dcvc_last_error());
    EXPECT(dcvc_rans_enc_flush(e) == 0, "flush");
    const uint8_t* p = nullptr;
    const int64_t n = dcvc_rans_enc_get_stream(e, &p);
    EXPECT(n >= 4, "stream of %lld bytes", (long long)n);
    return n > 0 ? std::vector<uint8_t>(p, p + n) : std::vector<uint8_t>();
}

inline bool is_escape(const Tables& t, int table, int symbol)
{
    const int value = symbol - t.offsets[table];
    return value < 0 || value >= t.sizes[table] - 2;
}

// Decodes a whole frame from `s`; returns the worst return code (0 or -4 allowed).  *same: every symbol equals the
// frame's.  *plain_differs: a symbol that was coded through its table (not an escape) came out different - such a stream
// can only end in the initial state by a 2^-23 accident, while an escape's raw bits are stored verbatim in the coder
// state: flipping one of them yields the VALID stream of a frame with another escaped value, which no rANS-level check
// can tell from the original (that takes a checksum, which the reference format has no room for).
int decode(dcvc_rans_dec* d, const std::vector<uint8_t>& s, const Frame& f, int zg, int yg, int z_start, bool* same,
           int* end_rc, const Tables* zt = nullptr, const Tables* yt = nullptr, bool* plain_differs = nullptr, bool two = false)
{
    int worst = 0;
    *same = true;
    if (plain_differs) *plain_differs = false;
    int rc = dcvc_rans_dec_set_stream(d, s.data(), (int64_t)s.size());
    if (rc != 0) {
        *same = false;
        *end_rc = rc;
        return rc;
    }
    rc = dcvc_rans_dec_decode_z(d, (int64_t)f.z.size(), zg, z_start, f.per_channel);
    EXPECT(rc == 0, "decode_z rc %d: %s", rc, dcvc_last_error());
    std::vector<int8_t> z(f.z.size());
    const int64_t got = dcvc_rans_dec_get(d, z.data(), (int64_t)z.size());
    if (got < 0) {
        EXPECT(got == -4, "get rc %lld", (long long)got);
        worst = (int)got;
        *same = false;
    } else {
        EXPECT(got == (int64_t)z.size(), "get returned %lld", (long long)got);
        *same = *same && std::memcmp(z.data(), f.z.data(), z.size()) == 0;
        if (plain_differs) {
            // table of element i as the coder picks it: the second coder's half starts at table start + n0 / per_channel
            // and counts its channels from its own first element (py_rans.cpp:53-61)
            const size_t n0 = two ? z.size() / 2 : z.size();
            for (size_t i = 0; i < z.size(); ++i) {
                const int table = i < n0 ? z_start + (int)(i / f.per_channel)
                                         : z_start + (int)(n0 / f.per_channel) + (int)((i - n0) / f.per_channel);
                if (z[i] != f.z[i] && !is_escape(*zt, table, f.z[i])) *plain_differs = true;
            }
        }
    }
    for (int h = 0; h < 2; ++h) {
        const size_t n = f.y[h].size();
        std::vector<uint8_t> idx(n);
        std::vector<int8_t> out(n, 77);
        for (size_t i = 0; i < n; ++i) idx[i] = (uint8_t)(f.y[h][i] & 0xff);
        if (h == 0) {       // asynchronous form, then the synchronous one
            rc = dcvc_rans_dec_decode_y(d, idx.data(), (int64_t)n, yg);
            EXPECT(rc == 0, "decode_y rc %d", rc);
            const int64_t g2 = dcvc_rans_dec_get(d, out.data(), (int64_t)n);
            rc = g2 < 0 ? (int)g2 : 0;
        } else if (g_compact_turn ^= 1) {
            // the compacted form of the same call (what the device hand-off of DMC / DMCI.decompress uses): kept indexes only, in
            // exact-size heap blocks, scattered back here for the comparison below
            std::vector<uint8_t> cidx;
            for (size_t i = 0; i < n; ++i)
                if (idx[i] != 0xFF) cidx.push_back(idx[i]);
            cidx.shrink_to_fit();
            std::vector<int8_t> co(cidx.size());
            rc = dcvc_rans_dec_decode_compact(d, cidx.data(), (int64_t)cidx.size(), yg, co.data());
            size_t k = 0;
            for (size_t i = 0; i < n; ++i) out[i] = idx[i] != 0xFF ? co[k++] : (int8_t)0;
        } else {
            rc = dcvc_rans_dec_decode_and_get_y(d, idx.data(), (int64_t)n, yg, out.data());
        }
        if (rc != 0) {
            EXPECT(rc == -4, "y decode rc %d: %s", rc, dcvc_last_error());
            worst = rc;
            *same = false;
            continue;
        }
        for (size_t i = 0; i < n; ++i) {
            const int want = idx[i] == 0xFF ? 0 : (int8_t)(f.y[h][i] >> 8);
            if (out[i] != want) {
                *same = false;
                if (plain_differs && !is_escape(*yt, idx[i], want)) *plain_differs = true;
            }
        }
    }
    *end_rc = dcvc_rans_dec_check_end(d);
    return worst;
}

}  // namespace

int main(int argc, char** argv)
{
    const int rounds = argc > 1 ? std::atoi(argv[1]) : 24;
    long n_trunc = 0, n_flip = 0, n_caught_by_decode = 0, n_escape_only = 0;
    for (int round = 0; round < rounds; ++round) {
        Rng r(1000 + round);
        const bool two = round & 1;
        const Tables zt = make_tables(r, 24, 19), yt = make_tables(r, 128, 3 + r.below(17));
        dcvc_rans_enc* e = dcvc_rans_enc_create();
        dcvc_rans_dec* d = dcvc_rans_dec_create();
        EXPECT(e && d, "create");
        const int yg_e = dcvc_rans_enc_add_cdf(e, yt.cdf.data(), yt.n, yt.stride, yt.sizes.data(), yt.offsets.data());
        const int zg_e = dcvc_rans_enc_add_cdf(e, zt.cdf.data(), zt.n, zt.stride, zt.sizes.data(), zt.offsets.data());
        const int yg = dcvc_rans_dec_add_cdf(d, yt.cdf.data(), yt.n, yt.stride, yt.sizes.data(), yt.offsets.data());
        const int zg = dcvc_rans_dec_add_cdf(d, zt.cdf.data(), zt.n, zt.stride, zt.sizes.data(), zt.offsets.data());
        EXPECT(yg == 0 && zg == 1 && yg_e == 0 && zg_e == 1, "add_cdf: %s", dcvc_last_error());
        dcvc_rans_enc_set_use_two(e, two);
        dcvc_rans_dec_set_use_two(d, two);

        const int zn = 8 + r.below(120), yn = round % 5 == 0 ? 3 : 40 + r.below(3000);
        Frame f = make_frame(r, zt, yt, zn, yn);
        const int channels = (zn + f.per_channel - 1) / f.per_channel;
        const int z_start = r.below(zt.n - channels + 1);
        const std::vector<uint8_t> s = encode(e, f, zg, yg, z_start);
        bool same = false;
        int end_rc = 0;

        // 1. intact stream: exact round trip, check_end passes
        int rc = decode(d, s, f, zg, yg, z_start, &same, &end_rc);
        EXPECT(rc == 0 && same, "round %d: intact stream does not round-trip (rc %d)", round, rc);
        EXPECT(end_rc == 0, "round %d: check_end rejects an intact stream: %s", round, dcvc_last_error());

        // 2. truncated at every length a container could hand over (all of the short ones, a sample of the rest)
        for (size_t cut = 0; cut < s.size(); cut += (cut < 24 || s.size() - cut < 24) ? 1 : 1 + r.below(97)) {
            std::vector<uint8_t> t(s.begin(), s.begin() + cut);
            t.shrink_to_fit();                  // exact-size heap block: ASan sees a read past the payload
            rc = decode(d, t, f, zg, yg, z_start, &same, &end_rc);
            EXPECT(rc == 0 || rc == -4 || (cut < 4 && rc == -1), "round %d cut %zu: rc %d", round, cut, rc);
            EXPECT(end_rc != 0, "round %d: truncation to %zu of %zu bytes not detected", round, cut, s.size());
            n_caught_by_decode += rc != 0;
            ++n_trunc;
        }
        // trailing garbage is not a valid stream either
        {
            std::vector<uint8_t> t(s);
            t.push_back(0x5a);      // (two coders: the appended byte becomes the second coder's first state byte)
            rc = decode(d, t, f, zg, yg, z_start, &same, &end_rc);
            EXPECT(end_rc != 0, "round %d: trailing byte not detected", round);
        }
        // 3. bit flips
        for (int k = 0; k < 200; ++k) {
            std::vector<uint8_t> t(s);
            const int nflip = 1 + r.below(3);
            for (int j = 0; j < nflip; ++j) t[(size_t)r.below((int)t.size())] ^= (uint8_t)(1u << r.below(8));
            if (t == s) continue;
            t.shrink_to_fit();
            bool plain = false;
            rc = decode(d, t, f, zg, yg, z_start, &same, &end_rc, &zt, &yt, &plain, two);
            EXPECT(rc == 0 || rc == -4, "round %d flip %d: rc %d", round, k, rc);
            EXPECT(end_rc != 0 || !plain, "round %d flip %d: wrong table-coded symbols accepted", round, k);
            n_escape_only += end_rc == 0 && !same;
            ++n_flip;
        }
        // 4. argument errors stay errors
        EXPECT(dcvc_rans_dec_set_stream(d, s.data(), 3) == -1, "short stream accepted");
        EXPECT(dcvc_rans_dec_set_stream(d, nullptr, 16) == -1, "null stream accepted");
        EXPECT(dcvc_rans_dec_set_stream(d, s.data(), (int64_t)s.size()) == 0, "set_stream");
        EXPECT(dcvc_rans_dec_decode_z(d, 10, 7, 0, 1) == -1, "unknown group accepted");
        EXPECT(dcvc_rans_dec_decode_z(d, 1000, zg, zt.n - 1, 1) == -1, "channels beyond the group accepted");
        {
            std::vector<uint8_t> idx(64, (uint8_t)(yt.n < 255 ? yt.n : 200));      // a table index past the group
            std::vector<int8_t> out(64);
            if (yt.n < 255) EXPECT(dcvc_rans_dec_decode_and_get_y(d, idx.data(), 64, yg, out.data()) == -1, "bad index accepted");
            std::vector<int16_t> sy(64, (int16_t)((1 << 8) | (yt.n < 255 ? yt.n : 200)));
            EXPECT(dcvc_rans_enc_reset(e) == 0, "reset");
            if (yt.n < 255) EXPECT(dcvc_rans_enc_encode_y(e, sy.data(), 64, yg) == -1, "encoder: bad index accepted");
        }
        dcvc_rans_enc_destroy(e);
        dcvc_rans_dec_destroy(d);
    }
    std::printf("rans_fuzz: %d rounds, %ld truncations (%ld already refused by a decode call), %ld bit-flip streams (%ld of them "
                "changed only the verbatim bits of escaped values = valid streams of another frame), %d failures\n",
                rounds, n_trunc, n_caught_by_decode, n_flip, n_escape_only, g_fail);
    return g_fail ? 1 : 0;
}
