// dcb_t128.hpp - DepthConvBlock tail for large maps, fp16: 128-pixel tiles, one 8-wave workgroup per CU (two waves per
// SIMD, 256 registers each), v_mfma_f32_32x32x16_f16, the FFN gate software-pipelined into each wave's own MFMA stream.
// Included by dcvc_nn.hip inside its anonymous namespace, after TailParams (same parameter block, plus `wt`).
// Further down, in the same form (waves = channel quarter x pixel half, weights as per-quarter fragment streams through a
// register ring): dcb_head128_kernel (the block's adaptor + first conv on large maps) and conv3x3_t128_kernel (3x3 stride-1
// convs as pixel tile x output-channel slice).
//
// Why this form (measurements: tools/coissue_mb.hip, tools/mb/coissue2_mb.hip, tools/mb/valu_rate_mb.hip):
//   * A SIMD has ONE issue port for vector and matrix instructions (~8 cycles per MFMA, ~3.8 per plain vector instruction,
//     ~8.5 per transcendental with two waves resident) beside the 32-cycle matrix pipe, and packed-fp32 vector instructions
//     (v_pk_*_f32) additionally take the matrix pipe itself (docs/experiments.md, round 4; round 2 read its own
//     micro-benchmark, whose vector wave hipcc had built from v_pk_fma_f32, as "two waves never overlap").  What a wave's
//     stream hides behind each of its 32x32x16 MFMAs is ~26 issue cycles of vector work.  So the gate g(u_lo) + g(u_hi)
//     of FFN chunk j is cut into pieces that are issued BETWEEN the MFMAs of W4 x v(j-1) and W3 x o -> u(j+1) of the same
//     wave (two named u accumulator sets, one barrier per chunk, v chunks double-buffered in LDS).  hipcc left alone
//     issues the gate as one block and sched_group_barrier pipelines of this length do not solve, so the kernel is written
//     as SLOTS (one MFMA + what is issued in its shadow) separated by __builtin_amdgcn_sched_barrier(0): the source order
//     is the schedule, while hipcc still allocates registers, counts waits and pads hazards (builtin MFMAs: an inline-asm
//     MFMA is opaque to it - its result registers are read too early, its operands overwritten too early).
//   * A single wave issues one plain vector instruction per ~6.8 cycles (transcendental ~9.8); two waves of a SIMD
//     together one per ~3.4.  The depthwise stage and the epilogues are pure vector work, and with one wave per SIMD
//     nobody covers an LDS / L2 round trip: hence two waves per SIMD.  Waves w and w + 4 own the same channels and split
//     the 128 pixels; they read the same weight fragments (the second read hits in L1).
//   * A weight fragment read from L2 feeds 128 pixels of one CU (the 64-pixel tiles at two workgroups per CU read every
//     fragment from L2 twice per 128 pixels, and their stream ran into the ~36 B/clk/CU the L2 -> L1 path delivers).
//   * The weights of the whole tail are packed per channel quarter in CONSUMPTION ORDER (W2, then W3 / W4 interleaved
//     chunk by chunk): one linear stream of 1 KiB fragments, read through a register ring D fragments deep that never
//     drains at a phase boundary (buffer loads: scalar descriptor + scalar running offset, no vector address arithmetic).
// Decomposition: wave w = (channel quarter cq = w & 3, pixel half ph = w >> 2).  It owns output channels
// [C/4 cq, C/4 (cq+1)) of the C-wide GEMMs (W2, W4) and, per FFN chunk of 64 v columns, the 16 columns [16 cq, +16), for its
// 64 pixels: its W3 tile has 32 rows = 16 u_lo rows | the 16 u_hi rows they pair with, so a pair lands in ONE lane
// (registers r and r + 8 of the 32x32 accumulator tile).
// Same store points / roundings as dcb_tail_kernel (d, W2 d + b2, o, v, r, out): the fp16 oracle covers both.
#pragma once

namespace t128 {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int TW = 16, TH = 8, M = TW * TH;   // 128 pixels per workgroup
constexpr int NW = 8, NTHR = NW * 64;
constexpr int PT = M / 32;                    // pixel tiles of 32 (MFMA N) per workgroup
constexpr int PTW = PT / 2;                   // ... per wave (its pixel half)
constexpr int PAD = 8;                        // LDS row pad (halfs): row stride = 4 dwords mod 8 -> conflict-free ds_read_b128
                                              // of the 32x32x16 B operand (16 distinct rows per lane group)
constexpr int VC = 64, LDV = VC + PAD;        // v columns per FFN chunk; row stride of a v chunk buffer
constexpr int HALO = (TH + 2) * (TW + 2);     // 180 pixels
constexpr int DW_SLAB = 64;
constexpr int LDS_S = DW_SLAB;                // halo slab rows unpadded: 8 lanes read one pixel's 128 bytes, the next 8 the next
                                              // pixel's - with 128-byte rows the four 64-byte pieces of a ds_read_b128 lane group
                                              // fall on the four bank quarters (a 144-byte stride made them collide two-way)

constexpr int PADF = 24;                      // dummy fragments behind every per-quarter stream: a ring (<= PADF deep) refills
                                              // without a clamp, and the two tile geometries share the packed streams

// Tile geometry of the tail kernel.  G128: the large-map form described above.  G32 (maps below 12 000 pixels, where 128-pixel
// tiles would leave most CUs idle): 32-pixel tiles, four waves = the four channel quarters (one per SIMD, up to two workgroups
// per CU), all 32 pixels in one MFMA column block.  A workgroup then streams the block's whole weights for 32 pixels, so the
// kernel lives on the stream: the ring is 24 fragments deep (tools/mb/stream_mb.hip: 255 workgroups read 2 MB each at
// 55 B/clk/CU with 64 KiB in flight per CU, 81 with 128 KiB; the 16x16x32 kernels this replaces got 20).
struct G128 {
    static constexpr int TW = t128::TW, TH = t128::TH, NW = t128::NW, D = 8, DW_SLAB = t128::DW_SLAB, WPS = 2;
};
struct G32 {
    static constexpr int TW = 8, TH = 4, NW = 4, D = 24 /* divides the fragments of two FFN steps at every width; 36 at C = 384 measured slower (ring in the accumulator file) */, DW_SLAB = 128, WPS = 1;
};
template <class G>
struct Geo : G {
    static constexpr int M = G::TW * G::TH, NTHR = G::NW * 64;
    static constexpr int PTW = (M / 32) / (G::NW / 4);         // pixel tiles of 32 per wave
    static constexpr int HALO = (G::TH + 2) * (G::TW + 2);
    static constexpr int LDS_S = G::DW_SLAB;                   // halo slab rows unpadded (see LDS_S above)
    static_assert(M % 32 == 0 && PTW >= 1 && G::NW % 4 == 0, "tile geometry");
};

// the weight streams of a width (what the host packs): independent of the tile geometry
template <int C>
struct Wcfg {
    static_assert(C % 64 == 0, "DepthConvBlock widths are padded to 64");
    static constexpr int NT = C / 32;                 // output channel tiles; tile cq + 4 i belongs to channel quarter cq
    static constexpr int NTW = (NT + 3) / 4;          // ... per channel quarter (C = 320: 3, 3, 2, 2 - the two missing
    static constexpr bool RAG = NT % 4 != 0;          // tiles run on zero weights and are dropped)
    static constexpr int KS = C / 16;                 // k-steps over C
    static constexpr int NCH = 2 * C / VC;            // FFN chunks
    static constexpr int F4 = 4 * NTW;                // fragments of one W4 chunk pass
    static constexpr int LDX = C + PAD;
    static constexpr int FRAGS = KS * NTW + NCH * KS + NCH * F4;   // fragments per channel quarter
    static constexpr int STREAM = FRAGS + PADF;       // + dummies: the ring refill never needs a clamp
};

template <int C, class G = G128, bool HEADIN = false>
struct Cfg : Wcfg<C> {
    using GE = Geo<G>;
    using WC = Wcfg<C>;
    using WC::NT; using WC::NTW; using WC::RAG; using WC::KS; using WC::NCH; using WC::F4; using WC::LDX; using WC::FRAGS; using WC::STREAM;
    static_assert(C % G::DW_SLAB == 0, "whole depthwise slabs");
    // ring depth: fragments requested ahead of their use.  Every phase's ring slots are static: phases start at a
    // compile-time offset into the ring
    static constexpr int D = G::D;
    static_assert(D <= PADF, "stream padding");
    // bufV: the two v chunk buffers of the FFN, earlier the halo slabs of the depthwise stage - two alternating ones, or (HEADIN)
    // every slab of the activation a computed in the kernel
    static constexpr size_t slab_elems = (size_t)(HEADIN && C / G::DW_SLAB > 2 ? C / G::DW_SLAB : 2) * GE::HALO * GE::LDS_S;
    static constexpr size_t v_elems = (size_t)2 * GE::M * LDV > slab_elems ? (size_t)2 * GE::M * LDV : slab_elems;
    // small tables staged in LDS once (every thread needs them, 8 - 64 threads each the same 16 bytes: through the L1 that
    // is 90 KB of requests per slab for the depthwise taps alone): depthwise taps [9][C] halfs, depthwise bias [C] floats,
    // FFN bias [4 C] floats
    static constexpr size_t TAB_BYTES = (size_t)9 * C * 2 + (size_t)C * 4 + (size_t)4 * C * 4;
    // HEADIN (the block's own first conv computed here on the tile + halo): the input rows of the halo pixels, padded to whole
    // 32-pixel MFMA column blocks, behind the tables
    static constexpr int HROWS = (GE::HALO + 31) / 32 * 32;
    static constexpr size_t HEAD_BYTES = HEADIN ? (size_t)HROWS * LDX * sizeof(half_t) : 0;
    static constexpr size_t LDS = ((size_t)GE::M * LDX + v_elems) * sizeof(half_t) + TAB_BYTES + HEAD_BYTES;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static_assert(NCH % 2 == 0 && NCH >= 4, "the chunk loop is unrolled by two");
    static_assert((2 * C) % VC == 0, "whole FFN chunks");
    // ring offsets (fragments consumed so far, mod D) at the start of: u(0), step 0, the odd / even steps of the loop,
    // the last step, the final W4 pass
    static constexpr int OFF_U0 = (KS * NTW) % D;
    static constexpr int OFF_S0 = (OFF_U0 + KS) % D;
    static constexpr int OFF_ODD = (OFF_S0 + KS) % D;
    static constexpr int OFF_EVEN = (OFF_ODD + F4 + KS) % D;
    static_assert((2 * (F4 + KS)) % D == 0, "the two-step loop body must turn the ring a whole number of times");
    static constexpr int OFF_LAST = OFF_ODD;
    static constexpr int OFF_FIN = (OFF_LAST + F4) % D;
};

__device__ __forceinline__ floatx16 mfma32(const half8& a, const half8& b, const floatx16& c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// C/D layout of the 32x32 tile: lane (pl = lane & 31 -> pixel, hh = lane >> 5), register reg -> row (channel)
// (reg & 3) + 8 (reg >> 2) + 4 hh: four quads of 4 consecutive channels at 8 g + 4 hh, g = 0..3.

// HEADIN (32-pixel tiles, block without adaptor and not fed by a fused head): a = gate(W1 x + b1) is computed here
// on the tile and its 1-pixel halo - 60 of 64 GEMM columns - from the block's input x (p.hx) and the W1 fragment stream the
// fused-head tails use (p.hwt), instead of by a head launch; pixels outside the picture get a = 0 (the depthwise conv's zero
// padding).  Same k order, bias, gate and fp16 rounding as the head kernels: the same `a`.
template <int C, class G = G128, bool HEADIN = false>
__global__ __launch_bounds__(Geo<G>::NTHR, G::WPS) void dcb_tail128_kernel(TailParams p)
{
    using TR = Traits<half_t>;
    using CF = Cfg<C, G, HEADIN>;
    using GE = Geo<G>;
    // (the geometry's values under the names the code below was written with: they shadow the namespace-level G128 constants)
    constexpr int TW = GE::TW, TH = GE::TH, M = GE::M, NTHR = GE::NTHR, PTW = GE::PTW, HALO = GE::HALO, DW_SLAB = GE::DW_SLAB,
                  LDS_S = GE::LDS_S;
    constexpr int NTW = CF::NTW, KS = CF::KS, NCH = CF::NCH, D = CF::D, LDX = CF::LDX, V = 8, GC = C / V;
    extern __shared__ __attribute__((aligned(32))) char smem[];
    half_t* bufX = reinterpret_cast<half_t*>(smem);
    half_t* bufV = bufX + M * LDX;
    half_t* tabW = bufV + CF::v_elems;                                   // [9][C] depthwise taps
    float* tabBd = reinterpret_cast<float*>(tabW + 9 * C);               // [C]
    float* tabB3 = tabBd + C;                                            // [4 C]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cqw = wave & 3, ph = wave >> 2;          // channel quarter, pixel half
    const int pl = lane & 31, hh = lane >> 5;
    const int tiles_x = (p.W + TW - 1) / TW;
    const int ty0 = (blockIdx.x / tiles_x) * TH, tx0 = (blockIdx.x % tiles_x) * TW;
    const half_t* a = reinterpret_cast<const half_t*>(p.a);
    const half_t* ident = reinterpret_cast<const half_t*>(p.ident);
    [[maybe_unused]] unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0, ts5 = 0, ts6 = 0;
    [[maybe_unused]] unsigned long long td[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    STAMP(ts0);

    // ---- weight stream of this wave's channel quarter: ring of D fragments, refilled right after use
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.wt)) + (size_t)cqw * CF::STREAM * 1024, 0, CF::STREAM * 1024, 0x00020000);
    const int wlane = lane * 16;
    int woff = 0;
    auto wload = [&]() __attribute__((always_inline)) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane, woff, 0);
        woff += 1024;
        return __builtin_bit_cast(half8, v);
    };
    half8 ring[D];

    // ---- the global reads of the prologue are requested up front: the activation tile + halo in 64-channel slabs
    // (registers), later the identity rows
    constexpr int HW_ = TW + 2, GS = DW_SLAB / V, NLD = (HALO * GS + NTHR - 1) / NTHR;
    constexpr int nslab = C / DW_SLAB;
    const half_t* wd = reinterpret_cast<const half_t*>(p.wd);
    const int dcs = (tid % GS) * V;                      // this thread's channel group inside a slab
    // Buffer loads with a range-checked descriptor: a halo pixel outside the picture gets an offset beyond the buffer and
    // reads zeros (the depthwise conv's padding).  No branch around a load - hipcc stops counting vmcnt behind one and
    // waits for EVERYTHING in flight, which would serialise the loads requested up front.
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    constexpr int OOB = 0x7FFFFFF0;
    const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<half_t*>(a), 0, (int)((long)p.H * p.W * p.lda * 2), 0x00020000);
    int hoff[NLD];       // byte offset of this thread's load k of a slab (the same for every slab)
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int hp = tid / GS + k * (NTHR / GS);
        const int y = ty0 - 1 + hp / HW_, x = tx0 - 1 + hp % HW_;
        hoff[k] = (hp < HALO && y >= 0 && y < p.H && x >= 0 && x < p.W) ? ((y * p.W + x) * (int)p.lda + dcs) * 2 : OOB;
#ifdef DCVC_DIAG
        if (p.ablate & 32) hoff[k] = (tid & 255) * 16;      // timing experiment: every load hits the same 4 KiB
#endif
    }
    Vec16 pre[nslab][NLD];
    auto fetch = [&](int slab) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NLD; ++k)
            pre[slab][k] = __builtin_bit_cast(Vec16, __builtin_amdgcn_raw_buffer_load_b128(arsrc, hoff[k], slab * DW_SLAB * 2, 0));
    };
    Vec16 wtap[9];
    floatx4 bd0, bd1;
    auto taps_fetch = [&](int slab) __attribute__((always_inline)) {
        const int c = slab * DW_SLAB + dcs;
#pragma unroll
        for (int t = 0; t < 9; ++t) wtap[t] = *reinterpret_cast<const Vec16*>(tabW + t * C + c);
        bd0 = *reinterpret_cast<const floatx4*>(tabBd + c);
        bd1 = *reinterpret_cast<const floatx4*>(tabBd + c + 4);
    };
    if constexpr (!HEADIN) fetch(0);
    // the small tables -> LDS (visible after the first barrier of the depthwise stage); requested behind slab 0, in
    // front of the other slabs: their LDS stores wait for nothing else
    {
        constexpr int NW16 = 9 * C * 2 / 16, NB16 = C * 4 / 16, N316 = 4 * C * 4 / 16;
        for (int i = tid; i < NW16 + NB16 + N316; i += NTHR) {
            const Vec16* src = i < NW16 ? reinterpret_cast<const Vec16*>(wd) + i
                               : i < NW16 + NB16 ? reinterpret_cast<const Vec16*>(p.bd) + (i - NW16)
                                                 : reinterpret_cast<const Vec16*>(p.b3) + (i - NW16 - NB16);
            reinterpret_cast<Vec16*>(tabW)[i] = *src;      // (the three tables are contiguous in LDS)
        }
    }
    if constexpr (!HEADIN) {
#pragma unroll
        for (int sl = 1; sl < nslab; ++sl) fetch(sl);
    }
    if constexpr (HEADIN) {
        // x on the tile + halo -> LDS rows (halo pixel order, zero beyond the picture / the halo), then W1 x on the matrix pipe:
        // this wave's channel tiles x the HROWS / 32 pixel blocks, + b1, gate, fp16 -> the (single) depthwise slab buffer
        constexpr int HROWS = CF::HROWS, HT = HROWS / 32, NLX = (HROWS * GC + NTHR - 1) / NTHR;
        half_t* xs = reinterpret_cast<half_t*>(reinterpret_cast<char*>(tabW) + CF::TAB_BYTES);
        const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<void*>(p.hx), 0, (int)((long)p.H * p.W * p.ldhx * 2), 0x00020000);
        Vec16 xv[NLX];
#pragma unroll
        for (int k = 0; k < NLX; ++k) {
            const int it = tid + k * NTHR, hp = it / GC, c = (it % GC) * V;
            const int y = ty0 - 1 + hp / HW_, x = tx0 - 1 + hp % HW_;
            const int off = (hp < HALO && y >= 0 && y < p.H && x >= 0 && x < p.W) ? ((y * p.W + x) * (int)p.ldhx + c) * 2 : OOB;
            xv[k] = __builtin_bit_cast(Vec16, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, off, 0, 0));
        }
        const __amdgpu_buffer_rsrc_t hrsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(p.hwt)) + (size_t)cqw * (KS * NTW + PADF) * 1024, 0,
            (KS * NTW + PADF) * 1024, 0x00020000);
        // W1's fragments of this wave (KS x NTW, k-step major) through the ring registers the tail's streams use later
        constexpr int NF = KS * NTW, HD = NF < D ? NF : D;
        int hoff = 0;
        auto hload = [&]() __attribute__((always_inline)) {
            typedef unsigned u32x4h __attribute__((ext_vector_type(4)));
            const u32x4h v = __builtin_amdgcn_raw_buffer_load_b128(hrsrc, wlane, hoff, 0);
            hoff += 1024;
            return __builtin_bit_cast(half8, v);
        };
#pragma unroll
        for (int k = 0; k < HD; ++k) ring[k] = hload();
#pragma unroll
        for (int k = 0; k < NLX; ++k) {
            const int it = tid + k * NTHR, hp = it / GC, c = (it % GC) * V;
            if (hp < HROWS) *reinterpret_cast<Vec16*>(xs + hp * LDX + c) = xv[k];
        }
        __syncthreads();
        floatx16 hacc[NTW][HT];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            half8 hb_[HT];
#pragma unroll
            for (int t = 0; t < HT; ++t) hb_[t] = *reinterpret_cast<const half8*>(xs + (32 * t + pl) * LDX + 8 * hh + 16 * s);
#pragma unroll
            for (int i = 0; i < NTW; ++i) {
                const int f = s * NTW + i, k = f % HD;
#pragma unroll
                for (int t = 0; t < HT; ++t) {
                    if (s == 0) {
                        floatx16 zero;
#pragma unroll
                        for (int r = 0; r < 16; ++r) zero[r] = 0.f;
                        hacc[i][t] = mfma32(ring[k], hb_[t], zero);
                    } else {
                        hacc[i][t] = mfma32(ring[k], hb_[t], hacc[i][t]);
                    }
                }
                if (f + HD < NF) ring[k] = hload();
            }
        }
        // a -> the slab buffers [slab][HALO][LDS_S] the depthwise stage reads (HEADIN keeps every slab, nothing is staged later)
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
            const int chb = 32 * (cqw + 4 * i) + 4 * hh;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const floatx4 bias = load_f4(p.hb1 + chb + 8 * g);
#pragma unroll
                for (int t = 0; t < HT; ++t) {
                    const int hp = 32 * t + pl;
                    const int y = ty0 - 1 + hp / HW_, x = tx0 - 1 + hp % HW_;
                    const bool inside = hp < HALO && y >= 0 && y < p.H && x >= 0 && x < p.W;
                    floatx4 v = {hacc[i][t][4 * g], hacc[i][t][4 * g + 1], hacc[i][t][4 * g + 2], hacc[i][t][4 * g + 3]};
                    v = v + bias;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = inside ? TR::gate(v[r]) : 0.f;
                    const int ch = chb + 8 * g;
                    if (hp < HALO) lds_store_quad<half_t>(bufV + (ch / DW_SLAB) * (HALO * LDS_S), LDS_S, hp, ch % DW_SLAB, v);
                }
            }
        }
    }
    constexpr int NID = M * GC / NTHR;
    // identity / output pass: item k of this thread -> pixel m = tid / 8 + (NTHR / 8) (k / G8), channel group tid % 8 + 8 (k % G8)
    // (eight threads cover 128 contiguous bytes of a pixel; no division per item, nothing to keep in registers)
    constexpr int G8 = C / 64, RPP = NTHR / 8;          // rows (pixels) per pass of the workgroup
    static_assert(M % RPP == 0 && NID == (M / RPP) * G8, "identity pass mapping");
    auto idmap = [&](int k, int& m, int& c) __attribute__((always_inline)) {
        c = ((tid & 7) + 8 * (k % G8)) * V;
        m = (tid >> 3) + RPP * (k / G8);
    };
    Vec16 idv[NID];
    const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<half_t*>(ident), 0, (int)((long)p.H * p.W * p.ldi * 2), 0x00020000);
    auto ident_fetch = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NID; ++k) {
            int m, icg;
            idmap(k, m, icg);
            const int y = ty0 + m / TW, x = tx0 + m % TW;
            int off = (y < p.H && x < p.W) ? ((y * p.W + x) * (int)p.ldi + icg) * 2 : OOB;
#ifdef DCVC_DIAG
            if (p.ablate & 32) off = (tid & 255) * 16;
#endif
            idv[k] = __builtin_bit_cast(Vec16, __builtin_amdgcn_raw_buffer_load_b128(irsrc, off, 0, 0));
        }
    };

    // LDS operand addresses of this lane: pixel row pl of this wave's pixel tile t (tile ph * PTW + t of the workgroup),
    // k offset 8 hh
    const int prow = (ph * PTW) * 32 + pl;
    const half_t* xb = bufX + prow * LDX + 8 * hh;
    auto bfrag_x = [&](int t, int s) __attribute__((always_inline)) {
        return *reinterpret_cast<const half8*>(xb + t * 32 * LDX + s * 16);
    };

    // W2 d, one MFMA per slot (runs behind the depthwise stage; its k-steps spread between that stage's multiply-adds were
    // built and measured slower - 45.8 against 45.0 us, docs/experiments.md)
    floatx16 acc[NTW][PTW];
    half8 g2b[PTW];
    auto gemm2_slot = [&](int s, int q) __attribute__((always_inline)) {      // MFMA q of k-step s of W2 d
        const int i = q / PTW, t = q % PTW, k = (s * NTW + i) % D;
        if (q == 0) {
#pragma unroll
            for (int tt = 0; tt < PTW; ++tt) g2b[tt] = bfrag_x(tt, s);
        }
        if (s == 0) {
            floatx16 zero;
#pragma unroll
            for (int r = 0; r < 16; ++r) zero[r] = 0.f;
            acc[i][t] = mfma32(ring[k], g2b[t], zero);
        } else {
            acc[i][t] = mfma32(ring[k], g2b[t], acc[i][t]);
        }
        if (t == PTW - 1) ring[k] = wload();
    };
    constexpr int G2Q = NTW * PTW;                      // MFMAs per k-step
    // ---- depthwise 3x3 (zero padding) + bias -> d in bufX.  A slab goes registers -> one of two LDS halo buffers
    // (bufV region) -> taps; one barrier per slab.
    {
        // a thread owns one channel group of two vertically adjacent pixels (y, y + 1 at x): 12 tap vectors instead of 18
        static_assert(NTHR / GS == M / 2 && TH % 2 == 0, "one pixel pair per thread and slab");
        const int pp = tid / GS, py = 2 * (pp / TW), px = pp % TW;      // pixel pair -> rows py, py + 1, column px of the tile
#pragma unroll
        for (int slab = 0; slab < nslab; ++slab) {
            half_t* hb = bufV + (HEADIN ? slab : (slab & 1)) * (HALO * LDS_S);
            if constexpr (!HEADIN) {
#pragma unroll
                for (int k = 0; k < NLD; ++k) {
                    const int hp = tid / GS + k * (NTHR / GS);
                    if (hp < HALO) *reinterpret_cast<Vec16*>(hb + hp * LDS_S + dcs) = pre[slab][k];
                }
            }
            if (slab == (nslab >= 2 ? nslab - 2 : 0)) ident_fetch();   // (registers of the first slabs are free again; two slabs of time to arrive - one at a single-slab width)
            if (slab == nslab - 1) {                // first turn of the weight ring: lands underneath this slab
#pragma unroll
                for (int k = 0; k < D; ++k) ring[k] = wload();
            }
            if (slab == 0) STAMP(td[0]);
            __syncthreads();
            if (slab < 4) STAMP(td[1 + slab]);
            taps_fetch(slab);
            const int c = slab * DW_SLAB + dcs;
            Vec16 tv[4][3];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
                    tv[r][kx] = *reinterpret_cast<const Vec16*>(hb + ((py + r) * HW_ + px + kx) * LDS_S + dcs);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                float sacc[V];
#pragma unroll
                for (int j = 0; j < V; ++j) sacc[j] = 0.f;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) fma_vec16<half_t>(tv[e + ky][kx], wtap[ky * 3 + kx], sacc);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    sacc[j] = sacc[j] + bd0[j];
                    sacc[4 + j] = sacc[4 + j] + bd1[j];
                }
                lds_store_vec<half_t>(bufX, LDX, (py + e) * TW + px, c, pack16<half_t>(sacc));
            }
        }
        __syncthreads();
    }
    STAMP(ts1);

    // output channel tile i of this wave = tile cqw + 4 i of the block (exists unless the width is ragged)
    auto tile_of = [&](int i) __attribute__((always_inline)) { return cqw + 4 * i; };
    auto tile_exists = [&](int i) __attribute__((always_inline)) { return !CF::RAG || cqw + 4 * i < CF::NT; };
    // epilogue biases of this lane's rows (quads 8 g + 4 hh of each channel tile): requested ahead of the GEMM they follow
    // where the registers allow (two tiles per wave)
    constexpr bool EB_PRE = NTW <= 2;
    floatx4 ebias[EB_PRE ? NTW : 1][4];
    auto ebias_load = [&](const float* b) __attribute__((always_inline)) {
        if constexpr (EB_PRE) {
#pragma unroll
            for (int i = 0; i < NTW; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) ebias[i][g] = load_f4(b + 32 * tile_of(i) + 4 * hh + 8 * g);
        }
    };
    auto ebias_get = [&](const float* b, int i, int g) __attribute__((always_inline)) {
        if constexpr (EB_PRE) return ebias[i][g];
        return load_f4(b + 32 * tile_of(i) + 4 * hh + 8 * g);
    };
    ebias_load(p.b2);

    // ---- W2 d
#pragma unroll
    for (int mm = 0; mm < KS * G2Q; ++mm) {
        gemm2_slot(mm / G2Q, mm % G2Q);
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- a C x C GEMM on the tile in bufX (this wave's channel quarter x its 64 pixels), weights through the ring in
    // slots: the pixel fragments of k-step s + 1 are requested during k-step s.  Used for the fused next-block head / 1x1
    // conv on r.  The ring must hold the stream's first D fragments; `next` supplies the following ones.
    auto gemm_c = [&](auto next) __attribute__((always_inline)) {
        half8 bc[PTW], bn[PTW];
#pragma unroll
        for (int t = 0; t < PTW; ++t) bc[t] = bfrag_x(t, 0);
        floatx16 zero;
#pragma unroll
        for (int r = 0; r < 16; ++r) zero[r] = 0.f;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
#pragma unroll
            for (int q = 0; q < NTW * PTW; ++q) {
                const int i = q / PTW, t = q % PTW, k = (s * NTW + i) % D;
                if (s + 1 < KS && q < PTW) bn[q] = bfrag_x(q, s + 1);
                acc[i][t] = mfma32(ring[k], bc[t], s == 0 ? zero : acc[i][t]);
                if (t == PTW - 1) ring[k] = next();
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int t = 0; t < PTW; ++t) bc[t] = bn[t];
        }
    };
    STAMP(ts2);
    __syncthreads();   // every wave has finished reading d
    // (W2 d + b2) -> fp16 -> bufX
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        if (!tile_exists(i)) continue;
        const int chb = 32 * tile_of(i) + 4 * hh;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const floatx4 bias = ebias_get(p.b2, i, g);
#pragma unroll
            for (int t = 0; t < PTW; ++t) {
                floatx4 v = {acc[i][t][4 * g], acc[i][t][4 * g + 1], acc[i][t][4 * g + 2], acc[i][t][4 * g + 3]};
                lds_store_quad<half_t>(bufX, LDX, prow + 32 * t, chb + 8 * g, v + bias);
            }
        }
    }
    __syncthreads();
    // o = (W2 d + b2) + x'.  A packed fp16 add IS the fp32 add + rounding of the other kernels: the exact sum of two fp16
    // numbers rounded to fp32 (24 >= 2 * 11 + 2 bits) and then to fp16 equals the sum rounded once.
#pragma unroll
    for (int k = 0; k < NID; ++k) {
        int m, icg;
        idmap(k, m, icg);
        const half8 o = __builtin_bit_cast(half8, lds_load_vec<half_t>(bufX, LDX, m, icg)) + __builtin_bit_cast(half8, idv[k]);
        lds_store_vec<half_t>(bufX, LDX, m, icg, __builtin_bit_cast(Vec16, o));
    }
    __syncthreads();

    STAMP(ts3);
    // ---- FFN, software-pipelined: step j = [W4 x v(j-1) -> acc] [W3 x o -> u(j+1)] with gate(u(j)) -> v(j) in between
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int t = 0; t < PTW; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.f;
    floatx16 ua[PTW], ub[PTW];
    // u accumulators start at the bias (lane's rows: quads 8 g + 4 hh of u_lo, then of u_hi): the first MFMA of a chain
    // takes the bias tuple as its C operand.  The bias of chunk j + 2 is requested during step j.
    const float* b3w = tabB3 + 16 * cqw + 4 * hh;       // (LDS copy)
    floatx16 biasv;
    auto bias_load = [&](int j) __attribute__((always_inline)) {
        j = j < NCH ? j : NCH - 1;
        const floatx4 l0 = load_f4(b3w + 64 * j), l1 = load_f4(b3w + 64 * j + 8);
        const floatx4 h0 = load_f4(b3w + 64 * j + 2 * C), h1 = load_f4(b3w + 64 * j + 2 * C + 8);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            biasv[r] = l0[r];
            biasv[4 + r] = l1[r];
            biasv[8 + r] = h0[r];
            biasv[12 + r] = h1[r];
        }
    };
    // The gate of a chunk = 4 PTW units (pixel tile t = n >> 2, quad g = (n >> 1) & 1, half hf = n & 1) of two pairs
    // -> v columns 16 cq + 8 g + 4 hh + 2 hf + {0, 1}, each unit cut into six pieces of 8 - 24 issue cycles that are
    // placed one per MFMA (48 pieces for the 48 MFMAs of a full step).  Same operations as Traits<half_t>::gate2.
    constexpr int NPIECE = 4 * PTW * 6;
    float ge[4], gr[4];
    floatx4 vq;
    auto gate_piece = [&](const floatx16 (&u)[PTW], half_t* vbuf, int i) __attribute__((always_inline)) {
        const int n = i / 6, phs = i % 6, t = n >> 2, g = (n >> 1) & 1, hf = n & 1;
        const float lo0 = u[t][4 * g + 2 * hf], lo1 = u[t][4 * g + 2 * hf + 1];
        const float hi0 = u[t][8 + 4 * g + 2 * hf], hi1 = u[t][8 + 4 * g + 2 * hf + 1];
        if (phs == 0) {
            ge[0] = __builtin_amdgcn_exp2f(lo0);
            ge[1] = __builtin_amdgcn_exp2f(lo1);
        } else if (phs == 1) {
            ge[2] = __builtin_amdgcn_exp2f(hi0);
            ge[3] = __builtin_amdgcn_exp2f(hi1);
        } else if (phs == 2) {
            gr[0] = __builtin_amdgcn_rcpf(1.0f + ge[0]);
            gr[1] = __builtin_amdgcn_rcpf(1.0f + ge[1]);
        } else if (phs == 3) {
            gr[2] = __builtin_amdgcn_rcpf(1.0f + ge[2]);
            gr[3] = __builtin_amdgcn_rcpf(1.0f + ge[3]);
        } else if (phs == 4) {
            vq[2 * hf] = DCVC_FMAF(hi0, gr[2], lo0 * gr[0]);
        } else {
            vq[2 * hf + 1] = DCVC_FMAF(hi1, gr[3], lo1 * gr[1]);
            if (hf == 1) lds_store_quad<half_t>(vbuf, LDV, prow + 32 * t, 16 * cqw + 8 * g + 4 * hh, vq);
        }
    };
    // One pipelined step (chunk j): the k-steps of [W4 x v(j-1) -> acc] (G4: 4 k-steps of NTW fragments) and of
    // [W3 x o -> u(j+1)] (G3: KS k-steps of one fragment) as a sequence of slots = one MFMA + what is issued in its shadow:
    // one LDS read of the next k-step's pixel fragments, the ring refill after a fragment's last MFMA, and (GATE) the gate
    // pieces of chunk j.  OFF: ring slot of the step's first fragment; jb: chunk whose bias is requested.
    auto wload_ffn = [&]() __attribute__((always_inline)) {
#ifdef DCVC_DIAG
        if (p.ablate & 256) {                       // timing experiment: the FFN re-reads the same fragment (one L1 line set)
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            return __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane, 0, 0));
        }
#endif
        return wload();
    };
    auto step = [&](auto G4c, auto G3c, auto GATEc, auto OFFc, const floatx16 (&ug)[PTW], floatx16 (&un)[PTW], half_t* vcur,
                    const half_t* vprev, int jb) __attribute__((always_inline)) {
        constexpr bool G4 = decltype(G4c)::value, G3 = decltype(G3c)::value, GATE = decltype(GATEc)::value;
        constexpr int OFF = decltype(OFFc)::value;
        constexpr int N4 = G4 ? 4 : 0, NS = N4 + (G3 ? KS : 0);
        constexpr int TM = N4 * NTW * PTW + (G3 ? KS * PTW : 0);       // MFMAs (slots) of the step
        constexpr int PPS = GATE ? (NPIECE + TM - 1) / TM : 0;          // gate pieces per slot
        auto load_b1 = [&](int idx, int t) __attribute__((always_inline)) {
#ifdef DCVC_DIAG
            if (p.ablate & 512) idx = idx < N4 ? 0 : N4;      // timing experiment: every k-step reads the same LDS fragment
#endif
            if (idx < N4) return *reinterpret_cast<const half8*>(vprev + (prow + 32 * t) * LDV + 8 * hh + idx * 16);
            return bfrag_x(t, idx - N4);
        };
        half8 bc[PTW], bn[PTW];
#pragma unroll
        for (int t = 0; t < PTW; ++t) bc[t] = load_b1(0, t);
        __builtin_amdgcn_sched_barrier(0);
        int m = 0;
#pragma unroll
        for (int idx = 0; idx < NS; ++idx) {
            const int nm = idx < N4 ? NTW * PTW : PTW;
#pragma unroll
            for (int q = 0; q < NTW * PTW; ++q) {
                if (q < nm) {
                    if (idx + 1 < NS && q < PTW) bn[q] = load_b1(idx + 1, q);
                    const int t = q % PTW;
                    if (idx < N4) {
                        const int i = q / PTW, k = (OFF + idx * NTW + i) % D;
                        acc[i][t] = mfma32(ring[k], bc[t], acc[i][t]);
                        if (t == PTW - 1) ring[k] = wload_ffn();
                    } else {
                        const int k = (OFF + N4 * NTW + (idx - N4)) % D;
                        un[t] = mfma32(ring[k], bc[t], idx == N4 ? biasv : un[t]);   // (first k-step: starts at the bias)
                        if (t == PTW - 1) ring[k] = wload_ffn();
                        if (idx == N4 && t == PTW - 1) bias_load(jb);                // the next chunk's, into the same registers
                    }
                    if constexpr (GATE) {
#ifdef DCVC_DIAG
                        if (!(p.ablate & 64))        // timing experiment: the FFN without its gate (wrong results)
#endif
#pragma unroll
                        for (int pp = 0; pp < PPS; ++pp)
                            if (m * PPS + pp < NPIECE) gate_piece(ug, vcur, m * PPS + pp);
                    }
                    ++m;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int t = 0; t < PTW; ++t) bc[t] = bn[t];
        }
#ifdef DCVC_DIAG
        if (p.ablate & 128) return;                 // timing experiment: no barrier between the chunks (wrong results)
#endif
        if constexpr (GATE) __syncthreads();
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    half_t* vA = bufV;
    half_t* vB = bufV + M * LDV;
#define T128_OFF(x) std::integral_constant<int, CF::x>{}
    bias_load(0);
    step(F_{}, T_{}, F_{}, T128_OFF(OFF_U0), ub, ua, vA, vB, 1);       // prologue: u(0)
    STAMP(ts4);
    step(F_{}, T_{}, T_{}, T128_OFF(OFF_S0), ua, ub, vA, vB, 2);
    for (int j = 1; j + 2 < NCH; j += 2) {
        step(T_{}, T_{}, T_{}, T128_OFF(OFF_ODD), ub, ua, vB, vA, j + 2);
        step(T_{}, T_{}, T_{}, T128_OFF(OFF_EVEN), ua, ub, vA, vB, j + 3);
    }
    step(T_{}, F_{}, T_{}, T128_OFF(OFF_LAST), ub, ua, vB, vA, 0);
    ebias_load(p.b4);
    step(T_{}, F_{}, F_{}, T128_OFF(OFF_FIN), ub, ua, vA, vB, 0);      // W4 x v(NCH-1)  (NCH - 1 is odd: its v is in buffer B = vprev)
#undef T128_OFF

    STAMP(ts5);
    // the fused next-block head / 1x1 conv (below): its first weight fragments are requested now, underneath the r epilogue
    const bool fused_next = p.nwt != nullptr;
    const __amdgpu_buffer_rsrc_t nrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(fused_next ? p.nwt : p.wt)) + (size_t)cqw * (KS * NTW + PADF) * 1024, 0,
        (KS * NTW + PADF) * 1024, 0x00020000);
    int noff = 0;
    auto nload = [&]() __attribute__((always_inline)) {
        typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
        const u32x4_ v = __builtin_amdgcn_raw_buffer_load_b128(nrsrc, wlane, noff, 0);
        noff += 1024;
        return __builtin_bit_cast(half8, v);
    };
    if (fused_next) {
#pragma unroll
        for (int k = 0; k < D; ++k) ring[k] = nload();
    }
    // ---- r = (W4 v + b4) + o, in place in bufX (each element is owned by one lane)
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        if (!tile_exists(i)) continue;
        const int chb = 32 * tile_of(i) + 4 * hh;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const floatx4 bias = ebias_get(p.b4, i, g);
#pragma unroll
            for (int t = 0; t < PTW; ++t) {
                const floatx4 o = lds_load_quad<half_t>(bufX, LDX, prow + 32 * t, chb + 8 * g);
                floatx4 v = {acc[i][t][4 * g], acc[i][t][4 * g + 1], acc[i][t][4 * g + 2], acc[i][t][4 * g + 3]};
                lds_store_quad<half_t>(bufX, LDX, prow + 32 * t, chb + 8 * g, (v + bias) + o);
            }
        }
    }
    __syncthreads();
    half_t* out = reinterpret_cast<half_t*>(p.out);
    if (out != nullptr) {
#pragma unroll
        for (int k = 0; k < NID; ++k) {
            int m, icg;
            idmap(k, m, icg);
            const int y = ty0 + m / TW, x = tx0 + m % TW;
            if (y < p.H && x < p.W) {
                const long pix = (long)y * p.W + x;
                float r[V];
                unpack16<half_t>(lds_load_vec<half_t>(bufX, LDX, m, icg), r);
                if (p.shortcut) {
                    float id[V];
                    unpack16<half_t>(*reinterpret_cast<const Vec16*>(ident + pix * p.ldi + icg), id);
#pragma unroll
                    for (int j = 0; j < V; ++j) r[j] = r[j] + id[j];
                }
                if (p.q != nullptr) {
#pragma unroll
                    for (int j = 0; j < V; ++j) r[j] = r[j] * ((icg + j) < p.c_log ? p.q[icg + j] : 1.0f);
                }
                *reinterpret_cast<Vec16*>(out + pix * p.ldo + icg) = pack16<half_t>(r);
            }
        }
    }
    STAMP(ts6);
#ifdef DCVC_DIAG
    if (p.stamps && tid == 0) {
        unsigned long long* o = p.stamps + (size_t)blockIdx.x * 16;
#pragma unroll
        for (int k = nslab + 1; k <= 4; ++k) td[k] = td[nslab < 4 ? nslab : 4];   // stamps of slabs this width does not have
        o[0] = ts1 - ts0;   // loads + depthwise
        o[1] = ts2 - ts1;   // GEMM2
        o[2] = ts3 - ts2;   // (+b2) store, o pass
        o[3] = ts4 - ts3;   // u(0)
        o[4] = ts5 - ts4;   // pipelined FFN
        o[5] = ts6 - ts5;   // r + store
        o[6] = ts0;
        o[7] = ts6;
        o[8] = td[0] - ts0;     // address setup + load issue + slab 0 staged
        o[9] = td[1] - td[0];   // first barrier
        o[10] = td[2] - td[1];  // slab 0 compute + slab 1 staging + barrier
        o[11] = td[3] - td[2];
        o[12] = td[4] - td[3];
        o[13] = ts1 - td[4];    // last slab compute + barrier
    }
#endif
    if (fused_next) {
        // Fused head of the next block (a' = gate(W1' r + b1')) or fused 1x1 conv ((W r + b) [* q]) on the tile still in bufX
        // (the host only fuses when this block has neither shortcut nor quant step: bufX holds exactly the values stored
        // above).  Same 32x32x16 GEMM as W2, the weights as one more fragment stream per channel quarter.  An f16 MFMA
        // accumulates its products in ascending k like a chain of fp32 additions whatever its shape, so the values equal
        // dcb_head_kernel's / conv_kernel's 16x16x32 ones bit for bit (tests: chained == unchained).
        gemm_c(nload);
        __syncthreads();   // every wave has finished reading r
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
            if (!tile_exists(i)) continue;
            const int chb = 32 * tile_of(i) + 4 * hh;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch0 = chb + 8 * g;
                const floatx4 bias = load_f4(p.nb1 + ch0);
                floatx4 qv = {1.f, 1.f, 1.f, 1.f};
                if (p.nplain && p.nq != nullptr) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) qv[r] = (ch0 + r) < p.n_log ? p.nq[ch0 + r] : 1.0f;
                }
#pragma unroll
                for (int t = 0; t < PTW; ++t) {
                    floatx4 v = {acc[i][t][4 * g], acc[i][t][4 * g + 1], acc[i][t][4 * g + 2], acc[i][t][4 * g + 3]};
                    v = v + bias;
                    if (p.nplain) {
                        if (p.nq != nullptr) v = v * qv;
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = TR::gate(v[r]);
                    }
                    lds_store_quad<half_t>(bufX, LDX, prow + 32 * t, ch0, v);
                }
            }
        }
        __syncthreads();
        half_t* na = reinterpret_cast<half_t*>(p.na_out);
#pragma unroll
        for (int k = 0; k < NID; ++k) {
            int m, icg;
            idmap(k, m, icg);
            const int y = ty0 + m / TW, x = tx0 + m % TW;
            if (y < p.H && x < p.W)
                *reinterpret_cast<Vec16*>(na + ((long)y * p.W + x) * p.nlda + icg) = lds_load_vec<half_t>(bufX, LDX, m, icg);
        }
    }
#ifdef DCVC_DIAG
    if (p.stamps && tid == 0) {
        unsigned long long ts7 = 0;
        STAMP(ts7);
        p.stamps[(size_t)blockIdx.x * 16 + 14] = ts7 - ts6;     // fused next-block head / conv
    }
#endif
}

// ------------------------------------------------------------------------------------------
// DepthConvBlock head for large maps, 128-pixel tiles (the dcb_head_kernel of dcvc_nn.hip in the form of the tail above):
//   x (one or two sources) -> [x' = Wa x + ba -> identity out] -> a = gate(W1 x' + b1)
// 8 waves = (channel quarter, pixel half), 32x32x16 MFMAs, each matrix as a per-quarter fragment stream through a register
// ring (wa128: Kin / 16 k-steps, source 0's channels first; w1128: the stream a fused-head tail uses).  The sources pass
// through LDS in chunks of up to 128 channels (the next chunk's loads are in flight during the current chunk's MFMAs), k
// ascending across chunks and sources: the accumulation order of the one-pass form, so the values equal dcb_head_kernel's
// bit for bit.  One workgroup per CU (a 1080p map has 255 tiles): 512 registers per lane.
// Geometry G32 (maps below 12 000 pixels, the heads in front of the 32-pixel tails): 32-pixel tiles, four waves.
template <int C, class G = G128>
struct HeadCfg {
    using CF = Wcfg<C>;
    static constexpr int DH = 4 * CF::NTW;          // ring depth = the fragments of 4 k-steps (64 input channels)
    static constexpr int KCH = 128;                 // channels per staged chunk
    static constexpr int LDS_S = KCH + PAD;
    static constexpr size_t LDS = ((size_t)Geo<G>::M * LDS_S + (size_t)Geo<G>::M * CF::LDX) * sizeof(half_t);
};

template <int C, bool ADAPT, class G = G128>
__global__ __launch_bounds__(Geo<G>::NTHR, 1) void dcb_head128_kernel(HeadParams p)
{
    using TR = Traits<half_t>;
    using CF = Wcfg<C>;
    using HC = HeadCfg<C, G>;
    using GE = Geo<G>;
    constexpr int TW = GE::TW, TH = GE::TH, M = GE::M, NTHR = GE::NTHR, PTW = GE::PTW;      // (shadow the G128 constants)
    constexpr int RPP = NTHR / 8, NPASS = M / RPP;          // pixels per staging pass of the workgroup (8 threads per pixel), passes
    static_assert(M % RPP == 0, "staging passes");
    constexpr int NTW = CF::NTW, KS = CF::KS, LDX = CF::LDX, V = 8, DH = HC::DH, KCH = HC::KCH, LDS_S = HC::LDS_S, G8 = C / 64;
    extern __shared__ __attribute__((aligned(32))) char smem[];
    half_t* bufY = reinterpret_cast<half_t*>(smem);        // x' (the W1 GEMM's operand), then the output tile
    half_t* bufS = bufY + M * LDX;                          // the staged source chunk
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cqw = wave & 3, ph = wave >> 2;
    const int pl = lane & 31, hh = lane >> 5;
    const int tiles_x = (p.W + TW - 1) / TW;
    const int ty0 = (blockIdx.x / tiles_x) * TH, tx0 = (blockIdx.x % tiles_x) * TW;
    const int prow = (ph * PTW) * 32 + pl;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    constexpr int OOB = 0x7FFFFFF0;
    auto tile_of = [&](int i) __attribute__((always_inline)) { return cqw + 4 * i; };
    auto tile_exists = [&](int i) __attribute__((always_inline)) { return !CF::RAG || cqw + 4 * i < CF::NT; };

    // the pixels this thread stages / stores (rows tid / 8 + RPP k of the tile): index in the picture or -1
    int pix2[NPASS];
#pragma unroll
    for (int k = 0; k < NPASS; ++k) {
        const int m = (tid >> 3) + RPP * k;
        const int y = ty0 + m / TW, x = tx0 + m % TW;
        pix2[k] = (y < p.H && x < p.W) ? y * p.W + x : -1;
    }
    // fragment streams: offsets in a vector register, so a read past the stream's end is range-checked to zero
    auto make_stream = [&](const void* base, int frags) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(base)) + (size_t)cqw * (frags + PADF) * 1024, 0, (frags + PADF) * 1024,
            0x00020000);
    };
    floatx16 acc[NTW][PTW];
    half8 ring[DH];
    auto zero_all = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NTW; ++i)
#pragma unroll
            for (int t = 0; t < PTW; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.f;
    };
    // acc += (64 nb input channels of the tile in X, row stride ldx) x (the stream's next 4 nb k-steps); the ring holds the
    // first DH fragments on entry and the following DH on exit
    auto gemm = [&](const half_t* X, int ldx, int nb, auto next) __attribute__((always_inline)) {
        const half_t* xb = X + prow * ldx + 8 * hh;
        half8 bc[PTW], bn[PTW];
#pragma unroll
        for (int t = 0; t < PTW; ++t) bc[t] = *reinterpret_cast<const half8*>(xb + t * 32 * ldx);
        __builtin_amdgcn_sched_barrier(0);
        for (int b = 0; b < nb; ++b) {
#pragma unroll
            for (int ss = 0; ss < 4; ++ss) {
                const int sn = (ss < 3 || b + 1 < nb) ? 4 * b + ss + 1 : 4 * b + ss;
#pragma unroll
                for (int q = 0; q < NTW * PTW; ++q) {
                    const int i = q / PTW, t = q % PTW, k = ss * NTW + i;
                    if (q < PTW) bn[q] = *reinterpret_cast<const half8*>(xb + q * 32 * ldx + sn * 16);
                    acc[i][t] = mfma32(ring[k], bc[t], acc[i][t]);
                    if (t == PTW - 1) ring[k] = next();
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int t = 0; t < PTW; ++t) bc[t] = bn[t];
            }
        }
    };
    // coalesced store of the [128][C] tile in bufY to a tensor with row stride ldg
    auto store_tile = [&](void* g, long ldg) __attribute__((always_inline)) {
        half_t* gp = reinterpret_cast<half_t*>(g);
#pragma unroll
        for (int k = 0; k < NPASS * G8; ++k) {
            const int m = (tid >> 3) + RPP * (k / G8), c = ((tid & 7) + 8 * (k % G8)) * V;
            const int px = pix2[k / G8];
            if (px >= 0) *reinterpret_cast<Vec16*>(gp + (long)px * ldg + c) = *reinterpret_cast<const Vec16*>(bufY + m * LDX + c);
        }
    };
    // acc + bias [-> gate] -> fp16 -> bufY
    auto epilogue = [&](const float* bias_p, bool gate) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
            if (!tile_exists(i)) continue;
            const int chb = 32 * tile_of(i) + 4 * hh;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const floatx4 bias = load_f4(bias_p + chb + 8 * g);
#pragma unroll
                for (int t = 0; t < PTW; ++t) {
                    floatx4 v = {acc[i][t][4 * g], acc[i][t][4 * g + 1], acc[i][t][4 * g + 2], acc[i][t][4 * g + 3]};
                    v = v + bias;
                    if (gate) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = TR::gate(v[r]);
                    }
                    lds_store_quad<half_t>(bufY, LDX, prow + 32 * t, chb + 8 * g, v);
                }
            }
        }
    };

    const __amdgpu_buffer_rsrc_t wrs = make_stream(p.w1128, KS * NTW);
    int woff = lane * 16;
    auto wnext = [&]() __attribute__((always_inline)) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrs, woff, 0, 0);
        woff += 1024;
        return __builtin_bit_cast(half8, v);
    };
    zero_all();
    if constexpr (ADAPT) {
        const int Kin = p.src.c0 + p.src.c1;
        const __amdgpu_buffer_rsrc_t ars = make_stream(p.wa128, (Kin / 16) * NTW);
        int aoff = lane * 16;
        auto anext = [&]() __attribute__((always_inline)) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ars, aoff, 0, 0);
            aoff += 1024;
            return __builtin_bit_cast(half8, v);
        };
        const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<void*>(p.src.x0), 0, (int)((long)p.H * p.W * p.src.ld0 * 2), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<void*>(p.src.c1 > 0 ? p.src.x1 : p.src.x0), 0, (int)((long)p.H * p.W * (p.src.c1 > 0 ? p.src.ld1 : p.src.ld0) * 2),
            0x00020000);
        const int n0 = (p.src.c0 + KCH - 1) / KCH, nchunk = n0 + (p.src.c1 + KCH - 1) / KCH;
        // chunk i: KCH channels (the last chunk of a source may hold 64) -> registers; returns its channel count
        u32x4 pre[2 * NPASS];
        auto fetch = [&](int i) __attribute__((always_inline)) {
            const bool s1 = i >= n0;
            const int cb = (s1 ? i - n0 : i) * KCH;
            const int left = (s1 ? p.src.c1 : p.src.c0) - cb, cnt = left < KCH ? left : KCH;
            const int ld = (int)(s1 ? p.src.ld1 : p.src.ld0);
#pragma unroll
            for (int k = 0; k < 2 * NPASS; ++k) {
                const int c = ((tid & 7) + 8 * (k & 1)) * V, px = pix2[k >> 1];
                const int off = (px >= 0 && c < cnt) ? (px * ld + cb + c) * 2 : OOB;
                pre[k] = s1 ? __builtin_amdgcn_raw_buffer_load_b128(rs1, off, 0, 0) : __builtin_amdgcn_raw_buffer_load_b128(rs0, off, 0, 0);
            }
            return cnt;
        };
        int cnt = fetch(0);
#pragma unroll
        for (int k = 0; k < DH; ++k) ring[k] = anext();
        for (int i = 0; i < nchunk; ++i) {
            if (i > 0) __syncthreads();      // every wave has finished reading the previous chunk
#pragma unroll
            for (int k = 0; k < 2 * NPASS; ++k)
                *reinterpret_cast<u32x4*>(bufS + ((tid >> 3) + RPP * (k >> 1)) * LDS_S + ((tid & 7) + 8 * (k & 1)) * V) = pre[k];
            const int cur = cnt;
            __syncthreads();
            if (i + 1 < nchunk) cnt = fetch(i + 1);
            gemm(bufS, LDS_S, cur / 64, anext);
        }
        // W1's first fragments are requested now; x' = Wa x + ba -> bufY, and out as the block's identity branch
#pragma unroll
        for (int k = 0; k < DH; ++k) ring[k] = wnext();
        epilogue(p.ba, false);
        __syncthreads();
        store_tile(p.ident, p.ldi);
        zero_all();
    } else {
        const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<void*>(p.src.x0), 0, (int)((long)p.H * p.W * p.src.ld0 * 2), 0x00020000);
        u32x4 xin[NPASS * G8];
#pragma unroll
        for (int k = 0; k < NPASS * G8; ++k) {
            const int c = ((tid & 7) + 8 * (k % G8)) * V, px = pix2[k / G8];
            xin[k] = __builtin_amdgcn_raw_buffer_load_b128(rs0, px >= 0 ? (px * (int)p.src.ld0 + c) * 2 : OOB, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < DH; ++k) ring[k] = wnext();
#pragma unroll
        for (int k = 0; k < NPASS * G8; ++k)
            *reinterpret_cast<u32x4*>(bufY + ((tid >> 3) + RPP * (k / G8)) * LDX + ((tid & 7) + 8 * (k % G8)) * V) = xin[k];
        __syncthreads();
    }
    gemm(bufY, LDX, KS / 4, wnext);
    __syncthreads();       // every wave has finished reading x': the output tile replaces it
    epilogue(p.b1, true);
    __syncthreads();
    store_tile(p.a_out, p.lda);
}

// ------------------------------------------------------------------------------------------
// 3x3 convolution (stride 1, zero padding 1, one source) as an implicit GEMM in the same form: a workgroup owns a 128-pixel
// tile x a slice of 256 output channels (blockIdx.y), the input tile + 1-pixel halo sits in LDS once and the nine taps read it
// at shifted rows; the slice's weights are one fragment stream per channel quarter in (tap, k-step, tile) order.  Written
// for decoder.up at 68x120 (128 -> 4 x 256, 2.4 MB of weights): with 32-pixel tiles every CU streamed all of them, here a
// CU streams one slice.  Accumulation order = conv_kernel's (tap major, k ascending): the same values bit for bit.
// Epilogue: + bias, plain or PixelShuffle(2) (a slice lies inside one sub-pixel phase: cs_p % 256 == 0).
struct Conv128Params {
    const void* x;
    long ldx;
    int H, W, kin;
    const void* wt;       // [slice][quarter][9 * kin / 16 * 2 + D] fragments of 1 KiB
    const float* b;
    void* out;
    long ldo;
    int shuffle, cs_p;
};

// NTW = channel tiles per wave: output slices of 128 NTW channels.  NTW = 2: one workgroup per CU (used when the grid then
// fits the chip in one round); NTW = 1: 128 registers, two workgroups per CU (finer work items for grids in between).
// The output tile replaces the input tile in LDS.
inline size_t conv128_lds(int kin, int ntw)
{
    const size_t in = (size_t)HALO * (kin + PAD), out = (size_t)M * (128 * ntw + PAD);
    return (in > out ? in : out) * sizeof(half_t);
}

template <int NTW>
__global__ __launch_bounds__(NTHR, NTW == 1 ? 4 : 2) void conv3x3_t128_kernel(Conv128Params p)
{
    constexpr int CONV_SLICE = 128 * NTW, DH = 4 * NTW, V = 8, LDO = CONV_SLICE + PAD, HW_ = TW + 2, G8 = CONV_SLICE / 64;
    extern __shared__ __attribute__((aligned(32))) char smem[];
    const int lds_s = p.kin + PAD;
    half_t* bufS = reinterpret_cast<half_t*>(smem);          // input tile + halo, rows lds_s apart
    half_t* bufO = bufS;                                      // ... then the output tile
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cqw = wave & 3, ph = wave >> 2;
    const int pl = lane & 31, hh = lane >> 5;
    const int tiles_x = (p.W + TW - 1) / TW;
    const int ty0 = (blockIdx.x / tiles_x) * TH, tx0 = (blockIdx.x % tiles_x) * TW;
    const int slice = blockIdx.y;
    const int prow = (ph * PTW) * 32 + pl;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    constexpr int OOB = 0x7FFFFFF0;
    const int ksteps = p.kin / 16, frags = 9 * ksteps * NTW;

    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.wt)) + ((size_t)slice * 4 + cqw) * (frags + DH) * 1024, 0, (frags + DH) * 1024,
        0x00020000);
    int woff = lane * 16;
    auto wnext = [&]() __attribute__((always_inline)) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrs, woff, 0, 0);
        woff += 1024;
        return __builtin_bit_cast(half8, v);
    };
    // input tile + halo -> LDS (pixels outside the picture read zeros: the padding)
    {
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<void*>(p.x), 0, (int)((long)p.H * p.W * p.ldx * 2), 0x00020000);
        const int G = p.kin / V, total = HALO * G;
        constexpr int U = 6;
        for (int it0 = tid; it0 < total; it0 += U * NTHR) {
            u32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int it = it0 + u * NTHR, hp = it / G, c = (it - hp * G) * V;
                const int y = ty0 - 1 + hp / HW_, x = tx0 - 1 + hp % HW_;
                const bool ok = it < total && y >= 0 && y < p.H && x >= 0 && x < p.W;
                v[u] = __builtin_amdgcn_raw_buffer_load_b128(xrs, ok ? ((y * p.W + x) * (int)p.ldx + c) * 2 : OOB, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int it = it0 + u * NTHR, hp = it / G, c = (it - hp * G) * V;
                if (it < total) *reinterpret_cast<u32x4*>(bufS + hp * lds_s + c) = v[u];
            }
        }
    }
    half8 ring[DH];
#pragma unroll
    for (int k = 0; k < DH; ++k) ring[k] = wnext();
    floatx16 acc[NTW][PTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int t = 0; t < PTW; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.f;
    __syncthreads();

    // B operand of pixel tile t at tap (ky, kx), k-step s: halo row ((py + ky) * 18 + px + kx), columns 16 s + 8 hh
    const half_t* xb[PTW];
#pragma unroll
    for (int t = 0; t < PTW; ++t) {
        const int m = prow + 32 * t;
        xb[t] = bufS + ((m / TW) * HW_ + m % TW) * lds_s + 8 * hh;
    }
    const int nb = p.kin / 64;                   // bodies of 4 k-steps per tap
    half8 bc[PTW], bn[PTW];
#pragma unroll
    for (int t = 0; t < PTW; ++t) bc[t] = *reinterpret_cast<const half8*>(xb[t]);
    __builtin_amdgcn_sched_barrier(0);
    for (int tap = 0; tap < 9; ++tap) {
        const int toff = ((tap / 3) * HW_ + tap % 3) * lds_s;
        const int tnext = (((tap + 1) / 3) * HW_ + (tap + 1) % 3) * lds_s;
        for (int b = 0; b < nb; ++b) {
#pragma unroll
            for (int ss = 0; ss < 4; ++ss) {
                // operand of the next k-step: the same tap, or the next tap's first (the very last step re-reads its own)
                int noff = toff + (4 * b + ss + 1) * 16;
                if (ss == 3 && b + 1 == nb) noff = tap < 8 ? tnext : toff + (4 * b + ss) * 16;
#pragma unroll
                for (int q = 0; q < NTW * PTW; ++q) {
                    const int i = q / PTW, t = q % PTW, k = ss * NTW + i;
                    if (q < PTW) bn[q] = *reinterpret_cast<const half8*>(xb[q] + noff);
                    acc[i][t] = mfma32(ring[k], bc[t], acc[i][t]);
                    if (t == PTW - 1) ring[k] = wnext();
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int t = 0; t < PTW; ++t) bc[t] = bn[t];
            }
        }
    }
    __syncthreads();      // every wave has finished reading the input tile
    // + bias -> fp16 -> output tile in LDS -> coalesced rows
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const int chb = 32 * (cqw + 4 * i) + 4 * hh;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const floatx4 bias = load_f4(p.b + slice * CONV_SLICE + chb + 8 * g);
#pragma unroll
            for (int t = 0; t < PTW; ++t) {
                floatx4 v = {acc[i][t][4 * g], acc[i][t][4 * g + 1], acc[i][t][4 * g + 2], acc[i][t][4 * g + 3]};
                lds_store_quad<half_t>(bufO, LDO, prow + 32 * t, chb + 8 * g, v + bias);
            }
        }
    }
    __syncthreads();
    half_t* out = reinterpret_cast<half_t*>(p.out);
    const int n0 = slice * CONV_SLICE;
    const int sp = p.shuffle ? n0 / p.cs_p : 0, sch0 = p.shuffle ? n0 - sp * p.cs_p : n0;
#pragma unroll
    for (int k = 0; k < 2 * G8; ++k) {
        const int m = (tid >> 3) + (M / 2) * (k / G8), c = ((tid & 7) + 8 * (k % G8)) * V;
        const int oy = ty0 + m / TW, ox = tx0 + m % TW;
        if (oy < p.H && ox < p.W) {
            const long pix = p.shuffle ? (long)(2 * oy + (sp >> 1)) * (2 * p.W) + (2 * ox + (sp & 1)) : (long)oy * p.W + ox;
            *reinterpret_cast<Vec16*>(out + pix * p.ldo + sch0 + c) = *reinterpret_cast<const Vec16*>(bufO + m * LDO + c);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Stride-2 convolutions (2x2 without padding, 3x3 with padding 1; one source) as an implicit GEMM on 32 output pixels
// (8 x 4) x 128 NTW output channels: four waves = the channel quarters, 32x32x16 MFMAs, the weights as one fragment stream per
// quarter in (tap, k-step, tile) order through a ring of 8 NTW fragments.  conv_kernel re-stages the input once per tap for a
// stride-2 conv (its one-pass form collides in the LDS banks: rows two pixels apart); here the input tile is staged once with
// the even and the odd input columns in separate planes, so the 32 pixels of a tap read consecutive rows.  An optional
// per-input-channel factor is applied while staging (x * in_q[c], rounded to fp16: conv_kernel's scale_vec).  Accumulation
// order = conv_kernel's (tap major, k ascending): the same values bit for bit.
struct ConvS2Params {
    const void* x;
    long ldx;
    int H, W, Ho, Wo, kin, k, pad;   // k = 2 or 3
    const void* wt;                   // [quarter][k k kin / 16 * NTW + D] fragments
    const float* b;
    const float* in_q;
    int in_qn;
    void* out;
    long ldo;
};

constexpr int S2_TW = 8, S2_TH = 4;
inline size_t conv_s2_lds(int kin, int k, int ntw)
{
    const int ih = 2 * S2_TH + k - 2, iwh = S2_TW + 1;
    const size_t in = (size_t)ih * 2 * iwh * (kin + PAD), out = (size_t)32 * (128 * ntw + PAD);
    return (in > out ? in : out) * sizeof(half_t);
}

template <int NTW>
__global__ __launch_bounds__(256, 1) void conv_s2_t32_kernel(ConvS2Params p)
{
    constexpr int N = 128 * NTW, DH = 8 * NTW, V = 8, LDO = N + PAD, TW_ = S2_TW, TH_ = S2_TH, NTHR_ = 256, IWH = S2_TW + 1, G8 = N / 64;
    extern __shared__ __attribute__((aligned(32))) char smem[];
    const int lds_s = p.kin + PAD;
    half_t* bufS = reinterpret_cast<half_t*>(smem);          // input tile: [row][column parity][column / 2][kin + PAD]
    half_t* bufO = bufS;                                      // ... then the output tile
    const int tid = threadIdx.x, lane = tid & 63;
    const int cqw = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = lane & 31, hh = lane >> 5;
    const int tiles_x = (p.Wo + TW_ - 1) / TW_;
    const int ty0 = (blockIdx.x / tiles_x) * TH_, tx0 = (blockIdx.x % tiles_x) * TW_;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    constexpr int OOB = 0x7FFFFFF0;
    const int taps = p.k * p.k, ksteps = p.kin / 16, frags = taps * ksteps * NTW;
    const int IH = 2 * TH_ + p.k - 2, IW = 2 * TW_ + p.k - 2;

    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.wt)) + (size_t)cqw * (frags + DH) * 1024, 0, (frags + DH) * 1024, 0x00020000);
    int woff = lane * 16;
    auto wnext = [&]() __attribute__((always_inline)) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrs, woff, 0, 0);
        woff += 1024;
        return __builtin_bit_cast(half8, v);
    };
    half8 ring[DH];
#pragma unroll
    for (int k = 0; k < DH; ++k) ring[k] = wnext();
    {   // input tile -> LDS (outside the picture: zeros = the padding)
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<void*>(p.x), 0, (int)((long)p.H * p.W * p.ldx * 2), 0x00020000);
        const int G = p.kin / V, total = IH * IW * G;
        const int iy0 = 2 * ty0 - p.pad, ix0 = 2 * tx0 - p.pad;
        constexpr int U = 6;
        for (int it0 = tid; it0 < total; it0 += U * NTHR_) {
            u32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int it = it0 + u * NTHR_, hp = it / G, c = (it - hp * G) * V;
                const int y = iy0 + hp / IW, x = ix0 + hp % IW;
                const bool ok = it < total && y >= 0 && y < p.H && x >= 0 && x < p.W;
                v[u] = __builtin_amdgcn_raw_buffer_load_b128(xrs, ok ? ((y * p.W + x) * (int)p.ldx + c) * 2 : OOB, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int it = it0 + u * NTHR_, hp = it / G, c = (it - hp * G) * V;
                const int r = hp / IW, col = hp % IW;
                Vec16 w = __builtin_bit_cast(Vec16, v[u]);
                if (p.in_q != nullptr) w = scale_vec<half_t>(w, p.in_q, c, p.in_qn);
                if (it < total) *reinterpret_cast<Vec16*>(bufS + ((r * 2 + (col & 1)) * IWH + (col >> 1)) * lds_s + c) = w;
            }
        }
    }
    floatx16 acc[NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    __syncthreads();

    // B operand of output pixel (py, px) at tap (ky, kx), k-step s: row ((2 py + ky) * 2 + (kx & 1)) * IWH + px + (kx >> 1)
    const int py = pl / TW_, px = pl % TW_;
    const half_t* xb = bufS + ((2 * py) * 2 * IWH + px) * lds_s + 8 * hh;
    auto tap_off = [&](int tap) __attribute__((always_inline)) {
        const int ky = tap / p.k, kx = tap - ky * p.k;
        return ((ky * 2 + (kx & 1)) * IWH + (kx >> 1)) * lds_s;
    };
    const int nb = p.kin / 128;                  // bodies of 8 k-steps per tap
    half8 bc = *reinterpret_cast<const half8*>(xb + tap_off(0)), bn;
    __builtin_amdgcn_sched_barrier(0);
    for (int tap = 0; tap < taps; ++tap) {
        const int toff = tap_off(tap), tnext = tap + 1 < taps ? tap_off(tap + 1) : toff;
        for (int b = 0; b < nb; ++b) {
#pragma unroll
            for (int ss = 0; ss < 8; ++ss) {
                int noff = toff + (8 * b + ss + 1) * 16;
                if (ss == 7 && b + 1 == nb) noff = tap + 1 < taps ? tnext : toff + (8 * b + ss) * 16;
#pragma unroll
                for (int i = 0; i < NTW; ++i) {
                    if (i == 0) bn = *reinterpret_cast<const half8*>(xb + noff);
                    acc[i] = mfma32(ring[ss * NTW + i], bc, acc[i]);
                    ring[ss * NTW + i] = wnext();
                    __builtin_amdgcn_sched_barrier(0);
                }
                bc = bn;
            }
        }
    }
    __syncthreads();      // every wave has finished reading the input tile
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const int chb = 32 * (cqw + 4 * i) + 4 * hh;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const floatx4 bias = load_f4(p.b + chb + 8 * g);
            floatx4 v = {acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]};
            lds_store_quad<half_t>(bufO, LDO, pl, chb + 8 * g, v + bias);
        }
    }
    __syncthreads();
    half_t* out = reinterpret_cast<half_t*>(p.out);
#pragma unroll
    for (int k = 0; k < G8; ++k) {
        const int m = tid >> 3, c = ((tid & 7) + 8 * k) * V;
        const int oy = ty0 + m / TW_, ox = tx0 + m % TW_;
        if (oy < p.Ho && ox < p.Wo)
            *reinterpret_cast<Vec16*>(out + ((long)oy * p.Wo + ox) * p.ldo + c) = *reinterpret_cast<const Vec16*>(bufO + m * LDO + c);
    }
}

}  // namespace t128
