// dcvc_nn.hip - the convolutional hot path of DCVC-RT on gfx950: fused DepthConvBlock
// (2 kernels instead of the reference's 8 launches, impl.cpp:53-121) and implicit-GEMM dense
// convolutions (1x1, 3x3 s1/s2, 2x2 s2) with fused bias / quant / PixelShuffle(2) / WSiLU epilogues.
// Decomposition and operand layouts: gemm_core.hpp.  C ABI: include/dcvc_amd.h.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "common.hpp"
#include "gemm_core.hpp"

namespace {

using dcvc::round_up;

// ------------------------------------------------------------------------------------------
// tile geometry: a workgroup owns M = 16*MT pixels arranged TH x TW
template <int MT>
struct Tile {
    static constexpr int M = 16 * MT;
    static constexpr int TW = MT >= 8 ? 16 : 8;
    static constexpr int TH = M / TW;
};

#ifndef DCVC_PF_SMALL
#define DCVC_PF_SMALL 4
#endif
template <typename T, int MT = 4>
struct Pf {   // weight prefetch depth (reduction groups in flight per wave)
    // 32-pixel tiles use a weight fragment for two MFMAs only: they are bound by latency x bytes in flight of the
    // weight stream (Little's law), not by the MFMA pipe - a deeper ring buys bandwidth there
    static constexpr int value = (sizeof(T) == 2 && MT == 2) ? DCVC_PF_SMALL : 2;
};

struct SrcPair {   // channel-concat of up to two HWC sources
    const void* x0;
    long ld0;
    int c0;
    const void* x1;
    long ld1;
    int c1;
};

// Loads channels [c, c+kVec) of the concat input at pixel index `pix` (or zeros if !valid).
template <typename T>
__device__ __forceinline__ Vec16 load_src(const SrcPair& s, long pix, int c, bool valid)
{
    Vec16 v = VEC16_ZERO;
    if (valid) {
        if (c < s.c0)
            v = *reinterpret_cast<const Vec16*>(reinterpret_cast<const T*>(s.x0) + pix * s.ld0 + c);
        else
            v = *reinterpret_cast<const Vec16*>(reinterpret_cast<const T*>(s.x1) + pix * s.ld1 + (c - s.c0));
    }
    return v;
}

__device__ __forceinline__ floatx4 load_f4(const float* p) { return *reinterpret_cast<const floatx4*>(p); }

// v[j] * q[c + j] for the kVec channels of a staged input vector, rounded to T like the stand-alone scale_channels kernel
// (channels beyond qn: unchanged).
template <typename T>
__device__ __forceinline__ Vec16 scale_vec(const Vec16& v, const float* __restrict__ q, int c, int qn)
{
    constexpr int V = Traits<T>::kVec;
    float f[V];
    unpack16<T>(v, f);
#pragma unroll
    for (int j = 0; j < V; ++j) f[j] = f[j] * ((c + j) < qn ? q[c + j] : 1.0f);
    return pack16<T>(f);
}

// ------------------------------------------------------------------------------------------
// DepthConvBlock, first kernel:  x -> [x' = adaptor(x)] -> a = wsilu(conv1(x') + b1)
struct HeadParams {
    SrcPair src;
    int H, W, C;   // C: physical channel count of the block
    const void* wa;
    const float* ba;
    void* ident;
    long ldi;
    const void* w1;
    const float* b1;
    void* a_out;
    long lda;
    int split;     // adaptor over two sources: stage and reduce one source at a time (half the LDS, same k order)
    const void* wa128;   // dcb_head128_kernel: adaptor / first conv as per-quarter fragment streams (dcb_t128.hpp)
    const void* w1128;
};

template <typename T, int MT, int NTW, bool ADAPT>
__global__ __launch_bounds__(NTHREADS) void dcb_head_kernel(HeadParams p)
{
    using TR = Traits<T>;
    using frag_t = typename TR::frag_t;
    constexpr int M = Tile<MT>::M, TW = Tile<MT>::TW, TH = Tile<MT>::TH, V = TR::kVec, PF = Pf<T, MT>::value;
    extern __shared__ __attribute__((aligned(32))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Kin = p.src.c0 + p.src.c1;
    const int C = p.C;
    const bool split = ADAPT && p.split;
    const int kstage = split ? (p.src.c0 > p.src.c1 ? p.src.c0 : p.src.c1) : Kin;   // channels staged at a time
    const int ldx = (kstage > C ? kstage : C) + TR::kPad;   // bufX is reused to stage the output tile
    const int ldy = C + TR::kPad;
    T* bufX = reinterpret_cast<T*>(smem);
    T* bufY = bufX + M * ldx;

    const int tiles_x = (p.W + TW - 1) / TW;
    const int ty0 = (blockIdx.x / tiles_x) * TH, tx0 = (blockIdx.x % tiles_x) * TW;

    // stage channels [cbeg, cbeg + cnt) of the concat input into columns [0, cnt) of bufX (coalesced 16-byte loads)
    auto stage = [&](int cbeg, int cnt) {
        const int G = cnt / V;
        for (int it = tid; it < M * G; it += NTHREADS) {
            const int m = it / G, c = (it - m * G) * V;
            const int y = ty0 + m / TW, x = tx0 + m % TW;
            const bool valid = (y < p.H) && (x < p.W);
            lds_store_vec<T>(bufX, ldx, m, c, load_src<T>(p.src, (long)y * p.W + x, cbeg + c, valid));
        }
    };
    stage(0, split ? p.src.c0 : Kin);
    __syncthreads();

    int tiles[NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i) tiles[i] = wave + NWAVE * i;
    const int pl = lane & 15, cq = (lane >> 4) * 4;
    floatx4 acc[MT][NTW];
    const int GC = C / V;

    if (ADAPT) {
        zero_acc(acc);
        if (split) {   // same reduction order as the one-pass form: source 0's channels, then source 1's
            gemm_acc<T, MT, NTW, PF>(acc, bufX, ldx, p.src.c0 / KG, reinterpret_cast<const frag_t*>(p.wa), Kin / KG, 0, tiles, lane);
            __syncthreads();
            stage(p.src.c0, p.src.c1);
            __syncthreads();
            gemm_acc<T, MT, NTW, PF>(acc, bufX, ldx, p.src.c1 / KG, reinterpret_cast<const frag_t*>(p.wa), Kin / KG, p.src.c0 / KG, tiles, lane);
        } else {
            gemm_acc<T, MT, NTW, PF>(acc, bufX, ldx, Kin / KG, reinterpret_cast<const frag_t*>(p.wa), Kin / KG, 0, tiles, lane);
        }
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
            const int ch0 = tiles[i] * 16 + cq;
            const floatx4 bias = load_f4(p.ba + ch0);
#pragma unroll
            for (int m = 0; m < MT; ++m) lds_store_quad<T>(bufY, ldy, m * 16 + pl, ch0, acc[m][i] + bias);
        }
        __syncthreads();
        // x' is the block's identity branch: write it out with coalesced 16-byte stores
        for (int it = tid; it < M * GC; it += NTHREADS) {
            const int m = it / GC, c = (it - m * GC) * V;
            const int y = ty0 + m / TW, x = tx0 + m % TW;
            if (y < p.H && x < p.W)
                *reinterpret_cast<Vec16*>(reinterpret_cast<T*>(p.ident) + ((long)y * p.W + x) * p.ldi + c) =
                    lds_load_vec<T>(bufY, ldy, m, c);
        }
    }

    zero_acc(acc);
    if (ADAPT)
        gemm_acc<T, MT, NTW, PF>(acc, bufY, ldy, C / KG, reinterpret_cast<const frag_t*>(p.w1), C / KG, 0, tiles, lane);
    else
        gemm_acc<T, MT, NTW, PF>(acc, bufX, ldx, Kin / KG, reinterpret_cast<const frag_t*>(p.w1), Kin / KG, 0, tiles, lane);
    if (!ADAPT) __syncthreads();   // bufX is still being read as the GEMM operand by other waves
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const int ch0 = tiles[i] * 16 + cq;
        const floatx4 bias = load_f4(p.b1 + ch0);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            floatx4 v = acc[m][i] + bias;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = TR::gate(v[r]);   // fp16 mode: kAct * wsilu, undone by the depthwise taps
            lds_store_quad<T>(bufX, ldx, m * 16 + pl, ch0, v);
        }
    }
    __syncthreads();
    for (int it = tid; it < M * GC; it += NTHREADS) {
        const int m = it / GC, c = (it - m * GC) * V;
        const int y = ty0 + m / TW, x = tx0 + m % TW;
        if (y < p.H && x < p.W)
            *reinterpret_cast<Vec16*>(reinterpret_cast<T*>(p.a_out) + ((long)y * p.W + x) * p.lda + c) =
                lds_load_vec<T>(bufX, ldx, m, c);
    }
}

// ------------------------------------------------------------------------------------------
// DepthConvBlock, second kernel:
//   d = dw3x3(a) + bd;  o = (W2 d + b2) + x';  v = wsilu(u_lo) + wsilu(u_hi), u = W3 o + b3;
//   out = ((W4 v + b4) + o) [+ x'] [* q]
struct TailParams {
    const void* a;
    long lda;
    const void* ident;
    long ldi;
    int H, W, C, c_log;
    const void* wd;      // [9][C] T
    const float* bd;
    const void* w2;
    const float* b2;
    const void* w3;
    const float* b3;     // [4C] : lo half | hi half
    const void* w4;
    const float* b4;
    int shortcut;
    const float* q;      // device, c_log entries, or NULL
    void* out;
    long ldo;
    // chained blocks: the NEXT block's first 1x1 conv + activation, computed on this block's output tile
    // while it is still in LDS (pointwise, so no halo is needed); NULL = not fused
    const void* nw1;
    const float* nb1;
    void* na_out;
    long nlda;
    // ... or a 1x1 convolution of the same width that consumes the block's output (decoder.conv2, the last conv of
    // y_prior_fusion / y_spatial_prior): nplain = 1 -> na_out = (W r + b) [* nq], no activation; out may then be NULL
    int nplain;
    const float* nq;
    int n_log;
    // HEADIN kernels (small maps, block without adaptor, one source): the block's own first conv + activation is
    // computed here on the tile + halo (hx = the block's input) instead of by dcb_head_kernel; `a` is then unused
    const void* hx;
    long ldhx;
    const void* hw1;
    const float* hb1;
    const void* hwt;     // dcb_tail128_kernel<..., HEADIN>: W1 as the per-quarter fragment stream (the one a fused-head tail reads)
    const void* wt;      // dcb_tail128_kernel: the tail's weights as per-quarter fragment streams (dcb_t128.hpp)
    const void* nwt;     // ... and the fused next-block head's / 1x1 conv's (the same matrix as nw1, 32x32x16 fragments)
    int ablate;          // debug: bit0 skip dw, bit1 skip GEMM2, bit2 skip FFN GEMM3, bit3 skip FFN GEMM4
    unsigned long long* stamps;   // diagnostic build only (DCVC_STAMPS): 8 cycle counters per workgroup
};

// Diagnostics (phase stamps, phase ablation) exist only in a -DDCVC_DIAG build (make diag): in the
// production kernel even a never-taken branch around a GEMM makes hipcc keep two copies of the
// accumulators and shuffle 64 registers per FFN chunk.
#ifdef DCVC_DIAG
#define ABLATED(bit) ((p.ablate & (bit)) != 0)
#define STAMP(var)                                                   \
    do {                                                             \
        if (p.stamps) {                                              \
            __builtin_amdgcn_sched_barrier(0);                       \
            var = __builtin_amdgcn_s_memtime();                      \
            __builtin_amdgcn_sched_barrier(0);                       \
        }                                                            \
    } while (0)
#else
#define ABLATED(bit) false
#define STAMP(var) \
    do {           \
    } while (0)
#endif

// channels per depthwise slab staged in LDS (with its 1-pixel halo): 128 in fp16 (widths that are
// multiples of 128), else 64
template <typename T>
__host__ __device__ constexpr int dw_slab_max() { return sizeof(T) == 2 ? 128 : 64; }

template <int MT, int NTW, int NW>
struct TailCfg {   // small blocks run two workgroups per CU: shallower prefetch, narrower FFN chunk.
                   // NW = waves per workgroup (4, or 8: more waves per SIMD hide latency by switching waves)
    static constexpr bool two_per_cu = (MT <= 4 && NTW * NW <= 16);
    static constexpr int NTV = (two_per_cu || NW == 8) ? 1 : 2;
    static constexpr int waves_per_simd = (two_per_cu ? 2 : 1) * NW / 4;
};

template <typename T, int MT, int NTV, int NW>
struct TailLds {
    using TR = Traits<T>;
    static constexpr int M = Tile<MT>::M;
    static constexpr int VC = NW * NTV * 16;
    static constexpr int HALO = (Tile<MT>::TH + 2) * (Tile<MT>::TW + 2);
    static constexpr int HPASS = (HALO + M - 1) / M;      // HEADIN: the halo tile as HPASS GEMM passes of M rows
    static constexpr int lds_slab = dw_slab_max<T>() + TR::kPad;
    static constexpr int ldv = VC + TR::kPad;
    static constexpr size_t v_elems = (size_t)M * ldv > (size_t)HALO * lds_slab ? (size_t)M * ldv : (size_t)HALO * lds_slab;
    static size_t bytes(int C, bool headin = false)
    {
        const size_t halo_elems = (size_t)HPASS * M * (C + TR::kPad);    // x, then a, on the tile + halo (aliases bufV)
        return ((size_t)M * (C + TR::kPad) + (headin && halo_elems > v_elems ? halo_elems : v_elems)) * sizeof(T);
    }
};

// RAG ("ragged"): the width is 4 channel tiles short of NTW * NW * 16 (C = 320 on 8 waves x 3 tiles): the
// waves whose last tile does not exist run it on a clamped weight tile and drop the result.
// HEADIN (32-pixel tiles, blocks without adaptor): the block's first conv + activation a = gate(W1 x + b1) is computed
// here on the tile and its 1-pixel halo (60 of 64 GEMM rows for a 4 x 8 tile) instead of by dcb_head_kernel - same k
// order, same epilogue, same fp16 rounding, so the same `a`; pixels outside the picture get a = 0 (the depthwise conv's
// zero padding).  One launch and one activation round trip less for the small-map blocks that follow a conv.
template <typename T, int MT, int NTW, int NW, bool RAG = false, bool HEADIN = false>
__global__ __launch_bounds__(NW * 64, (TailCfg<MT, NTW, NW>::waves_per_simd)) void dcb_tail_kernel(TailParams p)
{
    using TR = Traits<T>;
    using frag_t = typename TR::frag_t;
    constexpr int NTV = TailCfg<MT, NTW, NW>::NTV;
    constexpr int NTHREADS_ = NW * 64;    // v tiles per wave per chunk
    using LD = TailLds<T, MT, NTV, NW>;
    constexpr int M = Tile<MT>::M, TW = Tile<MT>::TW, TH = Tile<MT>::TH, V = TR::kVec;
    constexpr int PF = Pf<T, MT>::value;
    constexpr int PF3 = PF;   // GEMM3 has few MFMAs per k-group: look further ahead
    constexpr int VC = NW * NTV * 16;          // v columns per chunk
    static_assert(VC == LD::VC, "chunk width");
    extern __shared__ __attribute__((aligned(32))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = p.C;
    const int ldx = C + TR::kPad;
    constexpr int ldv = LD::ldv;
    T* bufX = reinterpret_cast<T*>(smem);
    T* bufV = bufX + M * ldx;

    const int tiles_x = (p.W + TW - 1) / TW;
    const int ty0 = (blockIdx.x / tiles_x) * TH, tx0 = (blockIdx.x % tiles_x) * TW;
    const T* a = reinterpret_cast<const T*>(p.a);
    const T* ident = reinterpret_cast<const T*>(p.ident);
    const int GC = C / V;
    [[maybe_unused]] unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0, tg3 = 0, tep = 0, tg4 = 0, tt = 0, tt2 = 0;
    STAMP(ts0);
    const int pl = lane & 15, cq = (lane >> 4) * 4;
    int tiles[NTW], wtiles[NTW];   // output tiles of this wave; wtiles = the same, clamped to existing weight tiles
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        tiles[i] = wave + NW * i;
        wtiles[i] = RAG ? min(tiles[i], C / 16 - 1) : tiles[i];
    }
    auto tile_exists = [&](int i) { return !RAG || tiles[i] < C / 16; };
    WPre<T, NTW, PF> pre2;   // W2's first groups are requested now and land during the depthwise stage
    if constexpr (!HEADIN)   // (HEADIN: after the head GEMM, whose own weight ring needs the registers)
        gemm_prefetch<T, NTW, PF>(pre2, C / KG, reinterpret_cast<const frag_t*>(p.w2), C / KG, 0, wtiles, lane);

    if constexpr (HEADIN) {
        // x on the tile + halo -> bufV (rows = halo pixels in (y, x) order, zero beyond the picture / the halo)
        constexpr int HW_ = TW + 2, HALO = LD::HALO, HROWS = LD::HPASS * M;
        const T* hx = reinterpret_cast<const T*>(p.hx);
        for (int it = tid; it < HROWS * GC; it += NTHREADS_) {
            const int hp = it / GC, c = (it - hp * GC) * V;
            const int y = ty0 - 1 + hp / HW_, x = tx0 - 1 + hp % HW_;
            Vec16 v = VEC16_ZERO;
            if (hp < HALO && y >= 0 && y < p.H && x >= 0 && x < p.W)
                v = *reinterpret_cast<const Vec16*>(hx + ((long)y * p.W + x) * p.ldhx + c);
            lds_store_vec<T>(bufV, ldx, hp, c, v);
        }
        __syncthreads();
        floatx4 hacc[LD::HPASS][MT][NTW];
#pragma unroll
        for (int ps = 0; ps < LD::HPASS; ++ps) {
            zero_acc(hacc[ps]);
            gemm_acc<T, MT, NTW, PF>(hacc[ps], bufV + ps * M * ldx, ldx, C / KG, reinterpret_cast<const frag_t*>(p.hw1), C / KG, 0,
                                     wtiles, lane);
        }
        __syncthreads();   // every wave has finished reading x: a replaces it in place
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
            const int ch0 = tiles[i] * 16 + cq;
            const floatx4 bias = load_f4(p.hb1 + wtiles[i] * 16 + cq);
            if (tile_exists(i)) {
#pragma unroll
                for (int ps = 0; ps < LD::HPASS; ++ps)
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        const int hp = ps * M + m * 16 + pl;
                        const int y = ty0 - 1 + hp / HW_, x = tx0 - 1 + hp % HW_;
                        const bool inside = hp < HALO && y >= 0 && y < p.H && x >= 0 && x < p.W;
                        floatx4 v = hacc[ps][m][i] + bias;
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = inside ? TR::gate(v[r]) : 0.f;
                        lds_store_quad<T>(bufV, ldx, hp, ch0, v);
                    }
            }
        }
        gemm_prefetch<T, NTW, PF>(pre2, C / KG, reinterpret_cast<const frag_t*>(p.w2), C / KG, 0, wtiles, lane);
        __syncthreads();
    }

    {   // depthwise 3x3 (zero padding), taps in (ky,kx) order, + bias.  The activation tile and its
        // 1-pixel halo go through LDS one 64-channel slab at a time (each input element is fetched
        // from L2 1.4-1.6x instead of 9x); the next slab is prefetched into registers meanwhile.
        // (HEADIN: the whole halo tile of `a` is in bufV already, rows ldx apart.)
        constexpr int DW_SLAB = (!RAG && dw_slab_max<T>() == 128 && (NTW * NW) % 8 == 0) ? 128 : 64;
        constexpr int HW_ = TW + 2, HALO = LD::HALO, GS = DW_SLAB / V, lds_s = LD::lds_slab;
        constexpr int NLD = (HALO * GS + NTHREADS_ - 1) / NTHREADS_;
        const T* wd = reinterpret_cast<const T*>(p.wd);
        const int nslab = C / DW_SLAB;
        const int tap_ld = HEADIN ? ldx : lds_s;       // row stride of the tap source
        Vec16 pre[HEADIN ? 1 : NLD];
        auto fetch = [&](int slab) {
            if constexpr (!HEADIN) {
#pragma unroll
                for (int k = 0; k < NLD; ++k) {
                    const int it = tid + k * NTHREADS_;
                    Vec16 v = VEC16_ZERO;
                    if (it < HALO * GS) {
                        const int hp = it / GS, c = slab * DW_SLAB + (it - hp * GS) * V;
                        const int y = ty0 - 1 + hp / HW_, x = tx0 - 1 + hp % HW_;
                        if (y >= 0 && y < p.H && x >= 0 && x < p.W && !ABLATED(1))
                            v = *reinterpret_cast<const Vec16*>(a + ((long)y * p.W + x) * p.lda + c);
                    }
                    pre[k] = v;
                }
            }
        };
        fetch(0);
        for (int slab = 0; slab < nslab; ++slab) {
            if constexpr (!HEADIN) {
#pragma unroll
                for (int k = 0; k < NLD; ++k) {
                    const int it = tid + k * NTHREADS_;
                    if (it < HALO * GS) {
                        const int hp = it / GS, cs = (it - hp * GS) * V;
                        *reinterpret_cast<Vec16*>(bufV + hp * lds_s + cs) = pre[k];   // natural order inside the slab
                    }
                }
            }
            // a thread always works on the same channel group of a slab (NTHREADS_ % GS == 0): its 9 tap
            // weights and the bias are fetched once per slab, before the barrier
            const int cs = (tid % GS) * V, c = slab * DW_SLAB + cs;
            const int tap_c = HEADIN ? c : cs;          // channel offset inside a tap source row
            Vec16 wtap[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) wtap[t] = *reinterpret_cast<const Vec16*>(wd + t * C + c);
            float bdv[V];
#pragma unroll
            for (int j = 0; j < V; j += 4) {
                const floatx4 b4 = load_f4(p.bd + c + j);
                bdv[j] = b4[0]; bdv[j + 1] = b4[1]; bdv[j + 2] = b4[2]; bdv[j + 3] = b4[3];
            }
            __syncthreads();
            if (slab + 1 < nslab) fetch(slab + 1);
            // a thread's pixels m0, m0 + MSTEP, ...: the nine tap vectors of the NEXT pixel are requested before the 72
            // multiply-adds of the current one (two named register sets, no copies), so the LDS latency is paid once per
            // slab instead of once per pixel
            constexpr int MSTEP = NTHREADS_ / GS, ITER = (M + MSTEP - 1) / MSTEP;
            constexpr bool GUARD = M % MSTEP != 0;    // (more threads than pixel x channel-group items: the rest idle)
            auto taps_load = [&](int m, Vec16 (&v)[9]) {
                const int my = m / TW, mx = m % TW;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
                        v[ky * 3 + kx] = *reinterpret_cast<const Vec16*>(bufV + ((my + ky) * HW_ + mx + kx) * tap_ld + tap_c);
            };
            auto taps_apply = [&](int m, const Vec16 (&v)[9]) {
                float s[V];
#pragma unroll
                for (int j = 0; j < V; ++j) s[j] = 0.f;
#pragma unroll
                for (int t = 0; t < 9; ++t) fma_vec16<T>(v[t], wtap[t], s);
#pragma unroll
                for (int j = 0; j < V; ++j) s[j] = s[j] + bdv[j];
                lds_store_vec<T>(bufX, ldx, m, c, pack16<T>(s));
            };
            const int m0 = tid / GS;
            auto live = [&](int m) { return !GUARD || m < M; };
            Vec16 ta[9], tb[9];
            if constexpr (MT < 4) {      // 32-pixel tiles keep the plain loop (the second register set would spill there)
#pragma unroll
                for (int it = 0; it < ITER; ++it) {
                    const int m = m0 + it * MSTEP;
                    if (live(m)) {
                        taps_load(m, ta);
                        taps_apply(m, ta);
                    }
                }
            } else {
            if (live(m0)) taps_load(m0, ta);
#pragma unroll
            for (int it = 0; it < ITER; it += 2) {
                const int ma = m0 + it * MSTEP, mb = ma + MSTEP, mc = mb + MSTEP;
                if (it + 1 < ITER && live(mb)) taps_load(mb, tb);
                if (live(ma)) taps_apply(ma, ta);
                if (it + 1 < ITER) {
                    if (it + 2 < ITER && live(mc)) taps_load(mc, ta);
                    if (live(mb)) taps_apply(mb, tb);
                }
            }
            }
            __syncthreads();
        }
    }

    STAMP(ts1);

    // identity rows for the o = ... + x' pass: requested before GEMM2 so they arrive underneath it
    constexpr int NID = (M * (16 * NW * NTW / V) + NTHREADS_ - 1) / NTHREADS_;
    Vec16 idv[NID];
#pragma unroll
    for (int k = 0; k < NID; ++k) {
        const int it = tid + k * NTHREADS_;
        const int m = it / GC, c = (it - m * GC) * V;
        const int y = ty0 + m / TW, x = tx0 + m % TW;
        idv[k] = VEC16_ZERO;
        if (it < M * GC && y < p.H && x < p.W)
            idv[k] = *reinterpret_cast<const Vec16*>(ident + ((long)y * p.W + x) * p.ldi + c);
    }

    floatx4 acc[MT][NTW];
    zero_acc(acc);
    if (!ABLATED(2))
        gemm_run<T, MT, NTW, PF>(acc, bufX, ldx, C / KG, pre2, reinterpret_cast<const frag_t*>(p.w2), C / KG, 0, wtiles, lane);
    const int vtiles = 2 * C / 16;
    int ut[2 * NTV];
#pragma unroll
    for (int j = 0; j < NTV; ++j) {
        ut[j] = wave * NTV + j;
        ut[j + NTV] = ut[j] + vtiles;
    }
    WPre<T, 2 * NTV, PF3> pre3;
    gemm_prefetch<T, 2 * NTV, PF3>(pre3, C / KG, reinterpret_cast<const frag_t*>(p.w3), C / KG, 0, ut, lane);
    __syncthreads();   // every wave has finished reading d
    STAMP(ts2);
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const int ch0 = tiles[i] * 16 + cq;
        const floatx4 bias = load_f4(p.b2 + wtiles[i] * 16 + cq);
        if (tile_exists(i)) {
#pragma unroll
            for (int m = 0; m < MT; ++m) lds_store_quad<T>(bufX, ldx, m * 16 + pl, ch0, acc[m][i] + bias);
        }
    }
    __syncthreads();
    // o = (W2 d + b2) + x'   (identity rows were loaded with coalesced 16-byte loads above)
#pragma unroll
    for (int k = 0; k < NID; ++k) {
        const int it = tid + k * NTHREADS_;
        if (it < M * GC) {
            const int m = it / GC, c = (it - m * GC) * V;
            float o[V], id[V];
            unpack16<T>(lds_load_vec<T>(bufX, ldx, m, c), o);
            unpack16<T>(idv[k], id);
#pragma unroll
            for (int j = 0; j < V; ++j) o[j] = o[j] + id[j];
            lds_store_vec<T>(bufX, ldx, m, c, pack16<T>(o));
        }
    }
    __syncthreads();

    STAMP(ts3);
    // FFN: C -> 4C -> chunk-add -> 2C -> C, the 4C-wide intermediate never leaves the CU
    zero_acc(acc);
    const int chunks = vtiles / (NW * NTV);
    floatx4 blo[NTV], bhi[NTV];
    if constexpr (TR::kActPrescaled) {
#pragma unroll
        for (int j = 0; j < NTV; ++j) {
            blo[j] = load_f4(p.b3 + ut[j] * 16 + cq);
            bhi[j] = load_f4(p.b3 + ut[j] * 16 + cq + 2 * C);
        }
    }
    for (int ch = 0; ch < chunks; ++ch) {
        floatx4 u[MT][2 * NTV];
        if constexpr (TR::kActPrescaled) {   // the bias seeds the accumulators; the next chunk's is requested now
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int j = 0; j < NTV; ++j) {
                    u[m][j] = blo[j];
                    u[m][j + NTV] = bhi[j];
                }
            const int adv = (ch + 1 < chunks) ? NW * NTV * 16 : 0;
#pragma unroll
            for (int j = 0; j < NTV; ++j) {
                blo[j] = load_f4(p.b3 + ut[j] * 16 + cq + adv);
                bhi[j] = load_f4(p.b3 + ut[j] * 16 + cq + 2 * C + adv);
            }
        } else {
            zero_acc(u);
#pragma unroll
            for (int j = 0; j < NTV; ++j) {   // biases requested before the GEMM that hides their latency
                blo[j] = load_f4(p.b3 + ut[j] * 16 + cq);
                bhi[j] = load_f4(p.b3 + ut[j] * 16 + cq + 2 * C);
            }
        }
        STAMP(tt);
        if (!ABLATED(4))
            gemm_run<T, MT, 2 * NTV, PF3>(u, bufX, ldx, C / KG, pre3, reinterpret_cast<const frag_t*>(p.w3), C / KG, 0, ut, lane);
        STAMP(tt2);
        tg3 += tt2 - tt;
        WPre<T, NTW, PF> pre4;   // W4's groups for this chunk arrive underneath the activation epilogue
        gemm_prefetch<T, NTW, PF>(pre4, VC / KG, reinterpret_cast<const frag_t*>(p.w4), 2 * C / KG, ch * (VC / KG), wtiles, lane);
#pragma unroll
        for (int j = 0; j < NTV; ++j) {
            const int lch0 = (wave * NTV + j) * 16 + cq;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                floatx4 lo = u[m][j], hi = u[m][j + NTV];
                if constexpr (!TR::kActPrescaled) {
                    lo = lo + blo[j];
                    hi = hi + bhi[j];
                }
                floatx4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = TR::gate2(lo[r], hi[r]);
                lds_store_quad<T>(bufV, ldv, m * 16 + pl, lch0, v);
            }
        }
        __syncthreads();
        STAMP(tt);
        tep += tt - tt2;
        if (!ABLATED(8))
            gemm_run<T, MT, NTW, PF>(acc, bufV, ldv, VC / KG, pre4, reinterpret_cast<const frag_t*>(p.w4), 2 * C / KG,
                                     ch * (VC / KG), wtiles, lane);
        if (ch + 1 < chunks) {   // next chunk's first W3 groups, requested before the barrier
#pragma unroll
            for (int j = 0; j < 2 * NTV; ++j) ut[j] += NW * NTV;
            gemm_prefetch<T, 2 * NTV, PF3>(pre3, C / KG, reinterpret_cast<const frag_t*>(p.w3), C / KG, 0, ut, lane);
        }
        __syncthreads();
        STAMP(tt2);
        tg4 += tt2 - tt;
    }
    STAMP(ts4);

    // r = (W4 v + b4) + o, accumulated in place in LDS (each element is owned by one lane)
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const int ch0 = tiles[i] * 16 + cq;
        const floatx4 bias = load_f4(p.b4 + wtiles[i] * 16 + cq);
        if (tile_exists(i)) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const floatx4 o = lds_load_quad<T>(bufX, ldx, m * 16 + pl, ch0);
                lds_store_quad<T>(bufX, ldx, m * 16 + pl, ch0, (acc[m][i] + bias) + o);
            }
        }
    }
    __syncthreads();
    T* out = reinterpret_cast<T*>(p.out);
    for (int it = tid; out != nullptr && it < M * GC; it += NTHREADS_) {     // (out == NULL: only the fused conv's output is wanted)
        const int m = it / GC, c = (it - m * GC) * V;
        const int y = ty0 + m / TW, x = tx0 + m % TW;
        if (y < p.H && x < p.W) {
            const long pix = (long)y * p.W + x;
            float r[V];
            unpack16<T>(lds_load_vec<T>(bufX, ldx, m, c), r);
            if (p.shortcut) {
                float id[V];
                unpack16<T>(*reinterpret_cast<const Vec16*>(ident + pix * p.ldi + c), id);
#pragma unroll
                for (int j = 0; j < V; ++j) r[j] = r[j] + id[j];
            }
            if (p.q != nullptr) {
#pragma unroll
                for (int j = 0; j < V; ++j) r[j] = r[j] * ((c + j) < p.c_log ? p.q[c + j] : 1.0f);
            }
            *reinterpret_cast<Vec16*>(out + pix * p.ldo + c) = pack16<T>(r);
        }
    }
    if (p.nw1 != nullptr) {
        // Fused head of the next DepthConvBlock: a' = gate(W1' r + b1') on the tile still in bufX (the host only
        // chains when this block has neither shortcut nor quant step, so bufX holds exactly the values stored
        // above).  Same k order and epilogue as dcb_head_kernel -> bit-identical to the unfused launch.
        floatx4 acc1[MT][NTW];
        zero_acc(acc1);
        gemm_acc<T, MT, NTW, PF>(acc1, bufX, ldx, C / KG, reinterpret_cast<const frag_t*>(p.nw1), C / KG, 0, wtiles, lane);
        __syncthreads();   // every wave has finished reading r
        if (p.nplain) {      // fused 1x1 conv: same epilogue as conv_kernel (bias, or bias * q), no activation
#pragma unroll
            for (int i = 0; i < NTW; ++i) {
                const int ch0 = tiles[i] * 16 + cq;
                const floatx4 bias = load_f4(p.nb1 + wtiles[i] * 16 + cq);
                floatx4 qv = {1.f, 1.f, 1.f, 1.f};
                if (p.nq != nullptr) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) qv[r] = (ch0 + r) < p.n_log ? p.nq[ch0 + r] : 1.0f;
                }
                if (tile_exists(i)) {
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        floatx4 v = acc1[m][i] + bias;
                        if (p.nq != nullptr) v = v * qv;
                        lds_store_quad<T>(bufX, ldx, m * 16 + pl, ch0, v);
                    }
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < NTW; ++i) {
                const int ch0 = tiles[i] * 16 + cq;
                const floatx4 bias = load_f4(p.nb1 + wtiles[i] * 16 + cq);
                if (tile_exists(i)) {
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        floatx4 v = acc1[m][i] + bias;
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = TR::gate(v[r]);
                        lds_store_quad<T>(bufX, ldx, m * 16 + pl, ch0, v);
                    }
                }
            }
        }
        __syncthreads();
        T* na = reinterpret_cast<T*>(p.na_out);
        for (int it = tid; it < M * GC; it += NTHREADS_) {
            const int m = it / GC, c = (it - m * GC) * V;
            const int y = ty0 + m / TW, x = tx0 + m % TW;
            if (y < p.H && x < p.W)
                *reinterpret_cast<Vec16*>(na + ((long)y * p.W + x) * p.nlda + c) = lds_load_vec<T>(bufX, ldx, m, c);
        }
    }
#ifdef DCVC_DIAG
    if (p.stamps) {
        unsigned long long te = 0;
        __syncthreads();
        STAMP(te);
        if (tid == 0) {
            unsigned long long* o = p.stamps + (size_t)blockIdx.x * 8;
            o[0] = ts1 - ts0;   // depthwise
            o[1] = ts2 - ts1;   // GEMM2 k-loop
            o[2] = ts3 - ts2;   // o = +b2 +x'
            o[3] = tg3;         // FFN GEMM3
            o[4] = tep;         // FFN wsilu/chunk-add + LDS store + barrier
            o[5] = tg4;         // FFN GEMM4
            o[6] = te - ts4;    // final epilogue + store
            o[7] = te - ts0;    // total
            if (p.ablate & 16) {   // diagnostic: absolute start / end stamps instead of two phase counters
                o[0] = ts0;
                o[1] = te;
            }
        }
    }
#endif
}

#include "dcb_t128.hpp"

// ------------------------------------------------------------------------------------------
// Dense convolution as implicit GEMM over (tap, cin)
struct ConvParams {
    SrcPair src;
    int H, W, Ho, Wo;
    int KH, KW, stride, pad;
    int N, n_log, cs_p;
    const void* w;
    const float* b;
    const float* q;
    int epi;
    void* out;
    long ldo;
    const float* in_q;   // optional per-input-channel factor applied while staging (x * in_q[c], rounded to T), in_qn entries
    int in_qn;
    int halo;      // KH x KW > 1: the input tile + halo is staged once and all taps read it (else: one tap at a time)
    int split_n;   // the passes over the output channels are spread over blockIdx.y (small maps) instead of looped
};

// HALO: KH x KW > 1 with the input tile + halo staged once (gemm_taps); a separate instantiation so that the 1x1
// kernels do not carry its registers (they dropped from two waves per SIMD to one when it was a runtime switch).
template <typename T, int MT, int NTW, bool HALO>
__global__ __launch_bounds__(NTHREADS) void conv_kernel(ConvParams p)
{
    using TR = Traits<T>;
    using frag_t = typename TR::frag_t;
    constexpr int M = Tile<MT>::M, TW = Tile<MT>::TW, TH = Tile<MT>::TH, PF = Pf<T>::value;   // (depth 4 measured slower here)
    extern __shared__ __attribute__((aligned(32))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Kin = p.src.c0 + p.src.c1;
    const int ldx = Kin + TR::kPad;
    T* bufX = reinterpret_cast<T*>(smem);
    const int tiles_x = (p.Wo + TW - 1) / TW;
    const int ty0 = (blockIdx.x / tiles_x) * TH, tx0 = (blockIdx.x % tiles_x) * TW;
    const int taps = p.KH * p.KW, kgin = Kin / KG, ntiles = p.N / 16;
    const int passes = (ntiles + NWAVE * NTW - 1) / (NWAVE * NTW);
    const int pl = lane & 15, cq = (lane >> 4) * 4;
    const int G = Kin / TR::kVec;
    T* out = reinterpret_cast<T*>(p.out);

    auto stage = [&](int tap) {
        const int ky = tap / p.KW, kx = tap - ky * p.KW;
        for (int it = tid; it < M * G; it += NTHREADS) {
            const int m = it / G, c = (it - m * G) * TR::kVec;
            const int oy = ty0 + m / TW, ox = tx0 + m % TW;
            const int iy = oy * p.stride - p.pad + ky, ix = ox * p.stride - p.pad + kx;
            const bool valid = (oy < p.Ho) && (ox < p.Wo) && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            Vec16 v = load_src<T>(p.src, (long)iy * p.W + ix, c, valid);
            if (p.in_q != nullptr) v = scale_vec<T>(v, p.in_q, c, p.in_qn);
            lds_store_vec<T>(bufX, ldx, m, c, v);
        }
    };

    [[maybe_unused]] const int HT_W = (TW - 1) * p.stride + p.KW, HT_H = (TH - 1) * p.stride + p.KH;
    [[maybe_unused]] const T* xrow[MT];
    if (taps == 1) {
        stage(0);
        __syncthreads();
    } else if constexpr (HALO) {
        const int iy0 = ty0 * p.stride - p.pad, ix0 = tx0 * p.stride - p.pad;
        const int total = HT_H * HT_W * G;
        constexpr int U = 8;               // loads in flight per thread (the tile is up to ~5 000 vectors)
        for (int it0 = tid; it0 < total; it0 += U * NTHREADS) {
            Vec16 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int it = it0 + u * NTHREADS;
                const int hp = it / G, c = (it - hp * G) * TR::kVec;
                const int iy = iy0 + hp / HT_W, ix = ix0 + hp % HT_W;
                const bool valid = it < total && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                v[u] = load_src<T>(p.src, (long)iy * p.W + ix, c, valid);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int it = it0 + u * NTHREADS;
                const int hp = it / G, c = (it - hp * G) * TR::kVec;
                if (p.in_q != nullptr) v[u] = scale_vec<T>(v[u], p.in_q, c, p.in_qn);
                if (it < total) lds_store_vec<T>(bufX, ldx, hp, c, v[u]);
            }
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int pix = m * 16 + (lane & 15);
            xrow[m] = bufX + ((pix / TW) * p.stride * HT_W + (pix % TW) * p.stride) * ldx + (lane >> 4) * 8;
        }
    }
    const int pass_lo = p.split_n ? (int)blockIdx.y : 0, pass_hi = p.split_n ? (int)blockIdx.y + 1 : passes;
    for (int pass = pass_lo; pass < pass_hi; ++pass) {
        int tiles[NTW];
        bool tvalid[NTW];
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
            const int t = pass * NWAVE * NTW + wave + NWAVE * i;
            tvalid[i] = t < ntiles;
            tiles[i] = tvalid[i] ? t : ntiles - 1;
        }
        floatx4 acc[MT][NTW];
        zero_acc(acc);
        if (taps == 1) {
            gemm_acc<T, MT, NTW, PF>(acc, bufX, ldx, kgin, reinterpret_cast<const frag_t*>(p.w), kgin, 0, tiles, lane);
        } else if constexpr (HALO) {
            gemm_taps<T, MT, NTW, PF>(acc, xrow, ldx, taps, p.KW, HT_W, kgin, reinterpret_cast<const frag_t*>(p.w), tiles, lane);
        } else {
            for (int tap = 0; tap < taps; ++tap) {
                __syncthreads();
                stage(tap);
                __syncthreads();
                gemm_acc<T, MT, NTW, PF>(acc, bufX, ldx, kgin, reinterpret_cast<const frag_t*>(p.w), taps * kgin,
                                         tap * kgin, tiles, lane);
            }
        }
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
            if (!tvalid[i]) continue;
            const int ch0 = tiles[i] * 16 + cq;
            const floatx4 bias = load_f4(p.b + ch0);
            floatx4 qv = {1.f, 1.f, 1.f, 1.f};
            if (p.epi == DCVC_EPI_BIAS_QUANT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) qv[r] = (ch0 + r) < p.n_log ? p.q[ch0 + r] : 1.0f;
            }
            int sdy = 0, sdx = 0, sch0 = ch0;
            if (p.epi == DCVC_EPI_SHUFFLE2) {
                const int s = ch0 / p.cs_p;
                sch0 = ch0 - s * p.cs_p;
                sdy = s >> 1;
                sdx = s & 1;
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int row = m * 16 + pl;
                const int oy = ty0 + row / TW, ox = tx0 + row % TW;
                if (oy >= p.Ho || ox >= p.Wo) continue;
                floatx4 v = acc[m][i] + bias;
                if (p.epi == DCVC_EPI_BIAS_QUANT) v = v * qv;
                if (p.epi == DCVC_EPI_WSILU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = TR::wsilu(v[r]);
                }
                if (p.epi == DCVC_EPI_SHUFFLE2)
                    global_store_quad<T>(out + ((long)(2 * oy + sdy) * (2 * p.Wo) + (2 * ox + sdx)) * p.ldo + sch0, v);
                else
                    global_store_quad<T>(out + ((long)oy * p.Wo + ox) * p.ldo + ch0, v);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// host side: weight packing and handles
struct DevBuf {
    void* p = nullptr;
    ~DevBuf()
    {
        if (p) (void)hipFree(p);
    }
    int upload(const void* src, size_t bytes)
    {
        if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) {
            p = nullptr;
            dcvc::set_error("hipMalloc(%zu) failed", bytes);
            return dcvc::E_MEM;
        }
        if (bytes && hipMemcpy(p, src, bytes, hipMemcpyHostToDevice) != hipSuccess) {
            dcvc::set_error("hipMemcpy H2D failed");
            return dcvc::E_HIP;
        }
        return 0;
    }
};

// get(n, k) -> logical weight (0 outside the logical range)
template <typename T>
int pack_upload(DevBuf& dst, int Np, int Kp, const std::function<float(int, int)>& get)
{
    const int kgs = Kp / KG, nt = Np / 16;
    std::vector<T> buf((size_t)nt * kgs * 64 * 8);
    for (int t = 0; t < nt; ++t)
        for (int g = 0; g < kgs; ++g)
            for (int lane = 0; lane < 64; ++lane) {
                const int r = lane & 15, q = lane >> 4;
                for (int j = 0; j < 8; ++j) {
                    const int k = g * KG + (sizeof(T) == 2 ? 8 * q + j : 4 * j + q);
                    buf[(((size_t)t * kgs + g) * 64 + lane) * 8 + j] = (T)get(t * 16 + r, k);
                }
            }
    return dst.upload(buf.data(), buf.size() * sizeof(T));
}

int pack_any(int dtype, DevBuf& dst, int Np, int Kp, const std::function<float(int, int)>& get)
{
    return dtype == DCVC_F16 ? pack_upload<half_t>(dst, Np, Kp, get) : pack_upload<float>(dst, Np, Kp, get);
}

int upload_f32(DevBuf& dst, int n, const std::function<float(int)>& get)
{
    std::vector<float> v(n);
    for (int i = 0; i < n; ++i) v[i] = get(i);
    return dst.upload(v.data(), v.size() * sizeof(float));
}

template <typename T>
int upload_T(DevBuf& dst, int n, const std::function<float(int)>& get)
{
    std::vector<T> v(n);
    for (int i = 0; i < n; ++i) v[i] = (T)get(i);
    return dst.upload(v.data(), v.size() * sizeof(T));
}


}  // namespace

struct dcvc_dcb {
    int dtype, cin, c, cin_p, c_p, shortcut, adapt;
    DevBuf wa, ba, w1, b1, wd, bd, w2, b2, w3, b3, w4, b4;
    DevBuf wa_t128; // fp16, widths 256 / 320 / 384: the adaptor as a fragment stream (dcb_head128_kernel)
    DevBuf w1_t128; // fp16, widths 256 / 320 / 384: W1 as a fragment stream (the previous block's tail computes this block's head)
    DevBuf wt128;   // fp16, widths 256 / 320 / 384: W2 | W3 | W4 once more as the fragment streams of dcb_tail128_kernel
};

struct dcvc_conv {
    int dtype, cin, cout, cin_p, n_p, cs_p, kh, kw, stride, pad, epi;
    DevBuf w, b;
    DevBuf w_t128;  // square 1x1 convs of width 256 / 320 / 384, fp16: the fragment stream for a fused launch behind a 128-pixel tail
    DevBuf w_c128[2];   // fp16, 3x3 s1 p1 convs: fragment streams of conv3x3_t128_kernel<1> (128-channel slices) and <2> (256)
    DevBuf w_s2;        // fp16, stride-2 convs (2x2 p0, 3x3 p1) with 128 / 256 output channels: the stream of conv_s2_t32_kernel
};

namespace {

template <typename T, int MT>
size_t head_lds(int kstage, int c, bool adapt)    // kstage: input channels staged at a time
{
    const int kx = kstage > c ? kstage : c;
    return (size_t)Tile<MT>::M * ((kx + Traits<T>::kPad) + (adapt ? c + Traits<T>::kPad : 0)) * sizeof(T);
}
template <typename T, int MT, int NTW, int NW>
size_t tail_lds(int c)
{
    return TailLds<T, MT, TailCfg<MT, NTW, NW>::NTV, NW>::bytes(c);
}

template <typename T, int MT, int NTW, int NW, bool RAG = false, bool HEADIN = false>
int launch_tail(const TailParams& tp, int grid, int C, hipStream_t st)
{
    const size_t lds = TailLds<T, MT, TailCfg<MT, NTW, NW>::NTV, NW>::bytes(C, HEADIN);
    int rc = set_lds(dcb_tail_kernel<T, MT, NTW, NW, RAG, HEADIN>, lds);
    if (rc) return rc;
    hipLaunchKernelGGL((dcb_tail_kernel<T, MT, NTW, NW, RAG, HEADIN>), dim3(grid), dim3(NW * 64), lds, st, tp);
    return 0;
}

template <typename K>
int set_lds(K kernel, size_t bytes)
{
    if (bytes > 160 * 1024) {
        dcvc::set_error("kernel needs %zu bytes of LDS (> 160 KiB)", bytes);
        return dcvc::E_ARG;
    }
    if (bytes > 64 * 1024) {   // raise the kernel's dynamic-LDS limit once (not per launch, not inside a graph capture)
        static std::mutex mu;
        static std::unordered_map<const void*, size_t> granted;
        const void* key = reinterpret_cast<const void*>(kernel);
        std::lock_guard<std::mutex> lk(mu);
        size_t& have = granted[key];
        if (have < bytes) {
            DCVC_HIP(hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
            have = bytes;
        }
    }
    return 0;
}


// ---- dcb_tail128_kernel: weight streams (one per wave, 1 KiB fragments of v_mfma_f32_32x32x16_f16's A operand in the
// order the wave consumes them) and launch
template <int C>
int pack_t128(DevBuf& dst, const std::function<float(int, int)>& W2, const std::function<float(int, int)>& W3,
              const std::function<float(int, int)>& W4)
{
    using CF = t128::Wcfg<C>;
    std::vector<half_t> buf((size_t)4 * CF::STREAM * 512, (half_t)0.f);
    for (int wave = 0; wave < 4; ++wave) {
        size_t f = 0;
        auto put = [&](const std::function<float(int, int)>& get, const std::function<int(int)>& row, int s) {
            half_t* o = &buf[((size_t)wave * CF::STREAM + f) * 512];
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) o[l * 8 + j] = (half_t)get(row(l & 31), 16 * s + 8 * (l >> 5) + j);
            ++f;
        };
        auto g3 = [&](int j) {   // u tile of chunk j: 16 u_lo rows | the 16 u_hi rows they pair with (physical halves 2 C wide)
            for (int s = 0; s < CF::KS; ++s)
                put(W3, [&](int r) { return (r < 16 ? 0 : 2 * C - 16) + 64 * j + 16 * wave + r; }, s);
        };
        auto g4 = [&](int j) {
            for (int s4 = 0; s4 < 4; ++s4)
                for (int i = 0; i < CF::NTW; ++i) put(W4, [&](int r) { return 32 * (wave + 4 * i) + r; }, 4 * j + s4);
        };
        for (int s = 0; s < CF::KS; ++s)
            for (int i = 0; i < CF::NTW; ++i) put(W2, [&](int r) { return 32 * (wave + 4 * i) + r; }, s);
        g3(0);
        for (int j = 0; j < CF::NCH; ++j) {
            if (j >= 1) g4(j - 1);
            if (j + 1 < CF::NCH) g3(j + 1);
        }
        g4(CF::NCH - 1);
        if (f != (size_t)CF::FRAGS) {
            dcvc::set_error("pack_t128: %zu fragments packed, %d expected", f, CF::FRAGS);
            return dcvc::E_ARG;
        }
    }
    return dst.upload(buf.data(), buf.size() * sizeof(half_t));
}

inline bool t128_width_ok(int c_p) { return c_p == 128 || c_p == 256 || c_p == 320 || c_p == 384 || c_p == 512; }   // (128, 512: 32-pixel tiles only)

// a C x C matrix (next block's first conv, a fused 1x1 conv) as the per-quarter fragment stream gemm_c reads
template <int C>
int pack_t128_square(DevBuf& dst, const std::function<float(int, int)>& W)
{
    using CF = t128::Wcfg<C>;
    constexpr int LEN = CF::KS * CF::NTW + t128::PADF;
    std::vector<half_t> buf((size_t)4 * LEN * 512, (half_t)0.f);
    for (int cq = 0; cq < 4; ++cq) {
        size_t f = 0;
        for (int s = 0; s < CF::KS; ++s)
            for (int i = 0; i < CF::NTW; ++i, ++f) {
                half_t* o = &buf[((size_t)cq * LEN + f) * 512];
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 8; ++j) o[l * 8 + j] = (half_t)W(32 * (cq + 4 * i) + (l & 31), 16 * s + 8 * (l >> 5) + j);
            }
    }
    return dst.upload(buf.data(), buf.size() * sizeof(half_t));
}

// a C x K matrix (the adaptor: K = padded input channels) in the same form: K / 16 k-steps
template <int C>
int pack_t128_rect(DevBuf& dst, int Kp, const std::function<float(int, int)>& W)
{
    using CF = t128::Wcfg<C>;
    const int ks = Kp / 16, LEN = ks * CF::NTW + t128::PADF;
    std::vector<half_t> buf((size_t)4 * LEN * 512, (half_t)0.f);
    for (int cq = 0; cq < 4; ++cq) {
        size_t f = 0;
        for (int s = 0; s < ks; ++s)
            for (int i = 0; i < CF::NTW; ++i, ++f) {
                half_t* o = &buf[((size_t)cq * LEN + f) * 512];
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 8; ++j) o[l * 8 + j] = (half_t)W(32 * (cq + 4 * i) + (l & 31), 16 * s + 8 * (l >> 5) + j);
            }
    }
    return dst.upload(buf.data(), buf.size() * sizeof(half_t));
}

// a conv with `taps` taps, Np / (128 ntw) slices x 4 quarters: fragments in (tap, k-step, tile) order, `padf` dummy fragments
// behind every quarter stream; W(n, tap, ci) = physical row n
inline int pack_t128_conv(DevBuf& dst, int ntw, int taps, int padf, int Np, int Kp, const std::function<float(int, int, int)>& W)
{
    const int slice = 128 * ntw;
    const int ks = Kp / 16, LEN = taps * ks * ntw + padf, nsl = Np / slice;
    std::vector<half_t> buf((size_t)nsl * 4 * LEN * 512, (half_t)0.f);
    for (int sl = 0; sl < nsl; ++sl)
        for (int cq = 0; cq < 4; ++cq) {
            size_t f = 0;
            for (int tap = 0; tap < taps; ++tap)
                for (int s = 0; s < ks; ++s)
                    for (int i = 0; i < ntw; ++i, ++f) {
                        half_t* o = &buf[(((size_t)sl * 4 + cq) * LEN + f) * 512];
                        for (int l = 0; l < 64; ++l)
                            for (int j = 0; j < 8; ++j)
                                o[l * 8 + j] = (half_t)W(sl * slice + 32 * (cq + 4 * i) + (l & 31), tap, 16 * s + 8 * (l >> 5) + j);
                    }
        }
    return dst.upload(buf.data(), buf.size() * sizeof(half_t));
}

inline int pack_t128_square_any(int c_p, DevBuf& dst, const std::function<float(int, int)>& W)
{
    return c_p == 128 ? pack_t128_square<128>(dst, W) : c_p == 256 ? pack_t128_square<256>(dst, W) : c_p == 320 ? pack_t128_square<320>(dst, W)
                      : c_p == 384 ? pack_t128_square<384>(dst, W) : pack_t128_square<512>(dst, W);
}

// DCVC_T128=0 keeps the 64-pixel tails on large maps (A/B measurements; both forms pass the same layer tests)
static bool t128_enabled()
{
    static const bool on = !(getenv("DCVC_T128") && atoi(getenv("DCVC_T128")) == 0);
    return on;
}

static bool t32_enabled()    // DCVC_T32=0: small-map tails (widths 256 / 384) by dcb_tail_kernel (A/B measurements, bit-identity checks)
{
    static const bool on = !(getenv("DCVC_T32") && atoi(getenv("DCVC_T32")) == 0);
    return on;
}

static bool t32_128_enabled()    // DCVC_T32_128=0: width-128 blocks by dcb_tail_kernel<..., HEADIN> (A/B measurements, bit-identity checks)
{
    static const bool on = !(getenv("DCVC_T32_128") && atoi(getenv("DCVC_T32_128")) == 0);
    return on;
}

static bool t32h_enabled()      // DCVC_T32H=0: width 256 keeps its separate head launch in front of the 32-pixel tail (A/B)
{
    static const bool on = !(getenv("DCVC_T32H") && atoi(getenv("DCVC_T32H")) == 0);
    return on;
}

static bool h128_enabled()   // DCVC_H128=0: large-map heads by dcb_head_kernel (A/B measurements, bit-identity checks)
{
    static const bool on = !(getenv("DCVC_H128") && atoi(getenv("DCVC_H128")) == 0);
    return on;
}

template <int C, class G = t128::G128, bool HEADIN = false>
int launch_tail128(const TailParams& tp, int H, int W, hipStream_t st)
{
    const int grid = ((H + G::TH - 1) / G::TH) * ((W + G::TW - 1) / G::TW);
    const size_t lds = t128::Cfg<C, G, HEADIN>::LDS;
    int rc = set_lds(t128::dcb_tail128_kernel<C, G, HEADIN>, lds);
    if (rc) return rc;
#ifdef DCVC_DIAG      // developer build only (make diag): in-kernel phase stamps, median over the workgroups
    static const bool want_stamps = getenv("DCVC_STAMPS") != nullptr;
    if (want_stamps) {
        TailParams q = tp;
        q.ablate = getenv("DCVC_ABLATE") ? atoi(getenv("DCVC_ABLATE")) : 0;     // timing experiments (wrong results)
        DCVC_HIP(hipMalloc(&q.stamps, (size_t)grid * 16 * sizeof(unsigned long long)));
        hipLaunchKernelGGL((t128::dcb_tail128_kernel<C, G, HEADIN>), dim3(grid), dim3(t128::Geo<G>::NTHR), lds, st, q);
        DCVC_HIP(hipStreamSynchronize(st));
        std::vector<unsigned long long> hs((size_t)grid * 16);
        DCVC_HIP(hipMemcpy(hs.data(), q.stamps, hs.size() * 8, hipMemcpyDeviceToHost));
        (void)hipFree(q.stamps);
        static int printed = 0;
        if (printed++ % 16 == 15) {
            const char* names[15] = {"loads+dw", "gemm2", "o_pass", "u0", "ffn", "r+store", "", "", "dw:issue+stage0", "barrier0", "slab0", "slab1", "slab2", "slab3", "fused"};
            fprintf(stderr, "[t128 stamps C=%d tile=%d grid=%d]", C, t128::Geo<G>::M, grid);
            constexpr int nslab = C / G::DW_SLAB;      // the kernel stamps min(nslab, 4) slabs: print only those
            for (int k = 0; k < 15; ++k) {
                if (k == 6 || k == 7 || (k >= 10 && k <= 13 && k - 10 >= nslab)) continue;
                std::vector<unsigned long long> v(grid);
                for (int b = 0; b < grid; ++b) v[b] = hs[(size_t)b * 16 + k];
                std::sort(v.begin(), v.end());
                fprintf(stderr, " %s=%llu", names[k], v[grid / 2]);
            }
            std::vector<unsigned long long> du(grid);
            for (int b = 0; b < grid; ++b) du[b] = hs[(size_t)b * 16 + 7] - hs[(size_t)b * 16 + 6];
            std::sort(du.begin(), du.end());
            fprintf(stderr, " | wg total min/med/max %llu %llu %llu\n", du[0], du[grid / 2], du[grid - 1]);
        }
        return 0;
    }
#endif
    hipLaunchKernelGGL((t128::dcb_tail128_kernel<C, G, HEADIN>), dim3(grid), dim3(t128::Geo<G>::NTHR), lds, st, tp);
    return 0;
}

// chained launches: this block's `a` lives in scratch slot a_slot (already there if head_done: the previous
// block's tail produced it); if next != NULL this block's tail also produces next's `a` in the other slot
struct ChainArgs {
    int head_done = 0, a_slot = 0;
    int separate_head = 0;      // measurement aid: never fold the head into the tail (so that `a` exists in the scratch)
    const dcvc_dcb* next = nullptr;
    // or: a 1x1 conv of the same width fused behind the block (its output replaces the block's)
    const dcvc_conv* conv = nullptr;
    const float* conv_q = nullptr;
    void* conv_out = nullptr;
    int64_t ldco = 0;
};

template <typename T, int MT, int NTW>
int launch_dcb(const dcvc_dcb* h, const SrcPair& src, int H, int W, const float* quant, void* out, int64_t ldo,
               void* scratch, hipStream_t st, hipEvent_t* ev = nullptr, ChainArgs ch = ChainArgs())
{
    const int C = h->c_p;
    const int grid = ((H + Tile<MT>::TH - 1) / Tile<MT>::TH) * ((W + Tile<MT>::TW - 1) / Tile<MT>::TW);
    const size_t P = (size_t)H * W;
    T* slots = reinterpret_cast<T*>(scratch);           // [a slot 0 | a slot 1 | x' of an adaptor block]
    T* a_buf = slots + (size_t)(ch.a_slot & 1) * P * C;
    T* a_next = slots + (size_t)((ch.a_slot & 1) ^ 1) * P * C;
    T* id_buf = slots + 2 * P * C;
    HeadParams hp{};
    hp.src = src;
    hp.H = H;
    hp.W = W;
    hp.C = C;
    hp.wa = h->wa.p;
    hp.ba = (const float*)h->ba.p;
    hp.ident = id_buf;
    hp.ldi = C;
    hp.w1 = h->w1.p;
    hp.b1 = (const float*)h->b1.p;
    hp.a_out = a_buf;
    hp.lda = C;
    const int kin = src.c0 + src.c1;
    // Small maps (32-pixel tiles, 4-wave tails), block without adaptor, one source: no head launch - the tail computes
    // `a` on its tile + halo itself (dcb_tail_kernel<..., HEADIN>)
    constexpr bool kHeadInKernel = sizeof(T) == 2 && MT == 2 && (NTW <= 4 || NTW == 6);   // (6: the 8-wave tail of C=384)
    // Small maps, widths 256 / 384 / 512 (fp16): the 32-pixel form of dcb_tail128_kernel (dcb_t128.hpp, geometry G32) - it wants `a`
    // from a head launch or the previous block's tail, like the large-map form
    bool t32 = false;
    if constexpr (sizeof(T) == 2 && MT == 2 && (NTW == 4 || NTW == 6 || NTW == 8)) {
        const bool fuse_ok = ch.next ? ch.next->w1_t128.p != nullptr : ch.conv ? ch.conv->w_t128.p != nullptr : true;
        t32 = h->wt128.p != nullptr && t128_enabled() && t32_enabled() && fuse_ok;
    }
    // ... with the block's first conv computed inside on the tile + halo (dcb_tail128_kernel<C, G32, HEADIN>) where the block has no
    // adaptor and its head was not computed by its predecessor: width 128 (the hyper path's blocks on 17x30 ... 68x120 maps, where
    // dcb_tail_kernel<..., HEADIN> ran before) and width 256 (one launch of 21.5 us instead of 6.1 + 17.5 us at 68x120; at width
    // 384 the head on the 60-pixel halo tile costs more than the head launch it saves: 41.9 against 8.0 + 32.1 us - not used)
    bool t32h = false;
    if constexpr (sizeof(T) == 2 && MT == 2 && (NTW == 2 || NTW == 4)) {
        const bool fuse_ok = ch.next ? ch.next->w1_t128.p != nullptr : ch.conv ? ch.conv->w_t128.p != nullptr : true;
        t32h = h->wt128.p != nullptr && h->w1_t128.p != nullptr && t128_enabled() && t32_enabled() && fuse_ok &&
               (NTW == 2 ? t32_128_enabled() : t32h_enabled()) && !ch.head_done && !ch.separate_head && !h->adapt && src.c1 == 0;
        if (t32h) t32 = false;
    }
    const bool head_in = kHeadInKernel && !ch.head_done && !ch.separate_head && !h->adapt && src.c1 == 0 && !t32 && !t32h;
    if (ev) DCVC_HIP(hipEventRecord(ev[0], st));
    // widths 256 / 320 / 384 on large maps, 256 / 384 / 512 in front of the 32-pixel tails: the head in the form of dcb_t128.hpp
    // (sources of 64-channel multiples)
    bool head128 = false;
    {
        constexpr bool kLarge = sizeof(T) == 2 && MT == 4 && (NTW == 4 || NTW == 5 || NTW == 6);
        constexpr bool kSmall = sizeof(T) == 2 && MT == 2 && (NTW == 4 || NTW == 6 || NTW == 8);
        if constexpr (kLarge || kSmall) {
            using G = typename std::conditional<kLarge, t128::G128, t128::G32>::type;
            head128 = t128_enabled() && h128_enabled() && (kLarge || t32) && !t32h && !ch.head_done && !head_in && h->w1_t128.p != nullptr &&
                      (h->adapt ? (h->wa_t128.p != nullptr && src.c0 % 64 == 0 && src.c1 % 64 == 0 && kin == h->cin_p)
                                : (src.c1 == 0 && src.c0 == NTW * 64));
            if (head128) {
                hp.wa128 = h->wa_t128.p;
                hp.w1128 = h->w1_t128.p;
                const int g128 = ((H + G::TH - 1) / G::TH) * ((W + G::TW - 1) / G::TW);
                const size_t lds = t128::HeadCfg<NTW * 64, G>::LDS;
                constexpr int NTHR_ = t128::Geo<G>::NTHR;
                int rc;
                if (h->adapt) {
                    rc = set_lds(t128::dcb_head128_kernel<NTW * 64, true, G>, lds);
                    if (rc) return rc;
                    hipLaunchKernelGGL((t128::dcb_head128_kernel<NTW * 64, true, G>), dim3(g128), dim3(NTHR_), lds, st, hp);
                } else {
                    rc = set_lds(t128::dcb_head128_kernel<NTW * 64, false, G>, lds);
                    if (rc) return rc;
                    hipLaunchKernelGGL((t128::dcb_head128_kernel<NTW * 64, false, G>), dim3(g128), dim3(NTHR_), lds, st, hp);
                }
            }
        }
    }
    if (ch.head_done || head_in || head128 || t32h) {
        // `a` was written by the previous block's tail / is computed by this block's tail / by the 128-pixel head
    } else if (h->adapt) {
        // two sources: one at a time through LDS (half the staging buffer: a 64-pixel tile of 256 + 256 channels then
        // leaves room for two workgroups per CU, i.e. half the weight fragments per pixel of the 32-pixel form)
        hp.split = src.c1 > 0;
        const int kstage = hp.split ? std::max(src.c0, src.c1) : kin;
        const size_t lds = head_lds<T, MT>(kstage, C, true);
        if (MT == 4 && lds > 80 * 1024) {
            // a 64-pixel tile of a wide two-source input leaves room for one workgroup per CU (two rounds over
            // a 136x240 map); 32-pixel tiles fit three per CU (97 -> 88 us for head + tail at 256+256 -> 256).
            // Head and tail tile shapes are independent.
            const int grid2 = ((H + Tile<2>::TH - 1) / Tile<2>::TH) * ((W + Tile<2>::TW - 1) / Tile<2>::TW);
            const size_t lds2 = head_lds<T, 2>(kstage, C, true);
            int rc = set_lds(dcb_head_kernel<T, 2, NTW, true>, lds2);
            if (rc) return rc;
            hipLaunchKernelGGL((dcb_head_kernel<T, 2, NTW, true>), dim3(grid2), dim3(NTHREADS), lds2, st, hp);
        } else {
            int rc = set_lds(dcb_head_kernel<T, MT, NTW, true>, lds);
            if (rc) return rc;
            hipLaunchKernelGGL((dcb_head_kernel<T, MT, NTW, true>), dim3(grid), dim3(NTHREADS), lds, st, hp);
        }
    } else {
        const size_t lds = head_lds<T, MT>(kin, C, false);
        int rc = set_lds(dcb_head_kernel<T, MT, NTW, false>, lds);
        if (rc) return rc;
        hipLaunchKernelGGL((dcb_head_kernel<T, MT, NTW, false>), dim3(grid), dim3(NTHREADS), lds, st, hp);
    }
    DCVC_LAUNCH_CHECK();
    if (ev) DCVC_HIP(hipEventRecord(ev[1], st));
    TailParams tp{};
    tp.a = a_buf;
    tp.lda = C;
    tp.ident = h->adapt ? (const void*)id_buf : src.x0;
    tp.ldi = h->adapt ? C : src.ld0;
    tp.H = H;
    tp.W = W;
    tp.C = C;
    tp.c_log = h->c;
    tp.wd = h->wd.p;
    tp.bd = (const float*)h->bd.p;
    tp.w2 = h->w2.p;
    tp.b2 = (const float*)h->b2.p;
    tp.w3 = h->w3.p;
    tp.b3 = (const float*)h->b3.p;
    tp.w4 = h->w4.p;
    tp.b4 = (const float*)h->b4.p;
    tp.shortcut = h->shortcut;
    tp.q = quant;
    tp.out = out;
    tp.ldo = ldo;
    if (head_in || t32h) {
        tp.hx = src.x0;
        tp.ldhx = src.ld0;
        tp.hw1 = h->w1.p;
        tp.hb1 = (const float*)h->b1.p;
        tp.hwt = h->w1_t128.p;
    }
    if (ch.next) {
        tp.nw1 = ch.next->w1.p;
        tp.nb1 = (const float*)ch.next->b1.p;
        tp.na_out = a_next;
        tp.nlda = C;
    } else if (ch.conv) {
        tp.nw1 = ch.conv->w.p;
        tp.nb1 = (const float*)ch.conv->b.p;
        tp.na_out = ch.conv_out;
        tp.nlda = ch.ldco;
        tp.nplain = 1;
        tp.nq = ch.conv_q;
        tp.n_log = ch.conv->cout;
    }
#ifdef DCVC_DIAG      // developer build only (make diag): phase ablation / in-kernel stamps
    {
        static const int abl = getenv("DCVC_ABLATE") ? atoi(getenv("DCVC_ABLATE")) : 0;
        tp.ablate = abl;
    }
    static const bool want_stamps = getenv("DCVC_STAMPS") != nullptr;
    unsigned long long* d_stamps = nullptr;
    if (want_stamps) {
        DCVC_HIP(hipMalloc(&d_stamps, (size_t)grid * 8 * sizeof(unsigned long long)));
        tp.stamps = d_stamps;
    }
#endif
    if constexpr (sizeof(T) == 2 && MT == 2 && (NTW == 2 || NTW == 4)) {
        if (t32h) {
            tp.wt = h->wt128.p;
            tp.nwt = ch.next ? ch.next->w1_t128.p : ch.conv ? ch.conv->w_t128.p : nullptr;
            int rc = launch_tail128<NTW * 64, t128::G32, true>(tp, H, W, st);
            if (rc) return rc;
            DCVC_LAUNCH_CHECK();
            if (ev) DCVC_HIP(hipEventRecord(ev[2], st));
            return 0;
        }
    }
    if constexpr (sizeof(T) == 2 && MT == 2 && (NTW == 4 || NTW == 6 || NTW == 8)) {
        if (t32) {
            tp.wt = h->wt128.p;
            tp.nwt = ch.next ? ch.next->w1_t128.p : ch.conv ? ch.conv->w_t128.p : nullptr;
            int rc = launch_tail128<NTW * 64, t128::G32>(tp, H, W, st);
            if (rc) return rc;
            DCVC_LAUNCH_CHECK();
            if (ev) DCVC_HIP(hipEventRecord(ev[2], st));
            return 0;
        }
    }
    if constexpr (sizeof(T) == 2 && MT == 4 && (NTW == 4 || NTW == 5 || NTW == 6)) {
        // large maps, widths 256 / 384: 128-pixel tiles, one 4-wave workgroup per CU, gate pipelined into the MFMA stream
        const void* nwt = ch.next ? ch.next->w1_t128.p : ch.conv ? ch.conv->w_t128.p : nullptr;
        if (h->wt128.p != nullptr && t128_enabled() && !head_in && (tp.nw1 == nullptr || nwt != nullptr)) {
            tp.wt = h->wt128.p;
            tp.nwt = nwt;
            int rc = launch_tail128<NTW * 64>(tp, H, W, st);
            if (rc) return rc;
            DCVC_LAUNCH_CHECK();
            if (ev) DCVC_HIP(hipEventRecord(ev[2], st));
            return 0;
        }
    }
    {
        // Widths of 384 and up (6+ channel tiles per wave) fit one workgroup per CU only.  Eight waves
        // (two per SIMD, half the channel tiles each, <= 256 VGPRs) give every SIMD a second wave of the
        // same tile to switch to: 144 -> 117 us at C=384, 136x240.  At C <= 256 two independent 4-wave
        // workgroups per CU do that job better (8 waves there measured 116 us vs 63 us; 128-pixel tiles on 8 waves,
        // MT = 8 x NTW = 2, spill at 256 VGPRs and measured 66 us against 59 us for the same build).
        int rc;
        if constexpr (NTW == 6 && kHeadInKernel)
            rc = head_in ? launch_tail<T, MT, 3, 8, false, true>(tp, grid, C, st) : launch_tail<T, MT, 3, 8>(tp, grid, C, st);
        else if constexpr (NTW >= 6 && NTW % 2 == 0 && sizeof(T) == 2)
            rc = launch_tail<T, MT, NTW / 2, 8>(tp, grid, C, st);
        else if constexpr (NTW == 5 && sizeof(T) == 2)
            rc = launch_tail<T, MT, 3, 8, true>(tp, grid, C, st);   // 20 tiles on 8 waves: 4 waves x 3 + 4 waves x 2
        else if constexpr (kHeadInKernel && NTW <= 4) {
            rc = head_in ? launch_tail<T, MT, NTW, 4, false, true>(tp, grid, C, st) : launch_tail<T, MT, NTW, 4>(tp, grid, C, st);
        } else
            rc = launch_tail<T, MT, NTW, 4>(tp, grid, C, st);
        if (rc) return rc;
    }
    DCVC_LAUNCH_CHECK();
    if (ev) DCVC_HIP(hipEventRecord(ev[2], st));
#ifdef DCVC_DIAG
    if (want_stamps) {   // diagnostic only: median cycles per phase over the workgroups
        DCVC_HIP(hipStreamSynchronize(st));
        std::vector<unsigned long long> hs((size_t)grid * 8);
        DCVC_HIP(hipMemcpy(hs.data(), d_stamps, hs.size() * 8, hipMemcpyDeviceToHost));
        (void)hipFree(d_stamps);
        static int printed = 0;
        if (printed++ % 16 == 15) {
            const char* names[8] = {"dw", "gemm2", "o_pass", "gemm3", "ffn_epi", "gemm4", "final", "total"};
            fprintf(stderr, "[stamps C=%d MT=%d grid=%d]", C, MT, grid);
            for (int k = 0; k < 8; ++k) {
                std::vector<unsigned long long> v(grid);
                for (int b = 0; b < grid; ++b) v[b] = hs[(size_t)b * 8 + k];
                std::sort(v.begin(), v.end());
                fprintf(stderr, " %s=%llu", names[k], v[grid / 2]);
            }
            fprintf(stderr, "\n");
            if (tp.ablate & 16) {
                unsigned long long t0 = ~0ull, t1 = 0;
                for (int b = 0; b < grid; ++b) {
                    t0 = std::min(t0, hs[(size_t)b * 8]);
                    t1 = std::max(t1, hs[(size_t)b * 8 + 1]);
                }
                std::vector<unsigned long long> st(grid), en(grid), du(grid);
                for (int b = 0; b < grid; ++b) {
                    st[b] = hs[(size_t)b * 8] - t0;
                    en[b] = hs[(size_t)b * 8 + 1] - t0;
                    du[b] = hs[(size_t)b * 8 + 7];
                }
                std::sort(st.begin(), st.end());
                std::sort(en.begin(), en.end());
                std::sort(du.begin(), du.end());
                fprintf(stderr, "[span %llu] start min/med/p90/max %llu %llu %llu %llu | end med/p90/max %llu %llu %llu | dur min/med/p90/max %llu %llu %llu %llu\n",
                        t1 - t0, st[0], st[grid / 2], st[grid * 9 / 10], st[grid - 1], en[grid / 2], en[grid * 9 / 10],
                        en[grid - 1], du[0], du[grid / 2], du[grid * 9 / 10], du[grid - 1]);
            }
        }
    }
#endif
    return 0;
}

template <typename T, int MT>
int dispatch_dcb(const dcvc_dcb* h, const SrcPair& src, int H, int W, const float* quant, void* out, int64_t ldo,
                 void* scratch, hipStream_t st, hipEvent_t* ev = nullptr, ChainArgs ch = ChainArgs())
{
    switch (h->c_p / 64) {
    case 1: return launch_dcb<T, MT, 1>(h, src, H, W, quant, out, ldo, scratch, st, ev, ch);
    case 2: return launch_dcb<T, MT, 2>(h, src, H, W, quant, out, ldo, scratch, st, ev, ch);
    case 3: return launch_dcb<T, MT, 3>(h, src, H, W, quant, out, ldo, scratch, st, ev, ch);
    case 4: return launch_dcb<T, MT, 4>(h, src, H, W, quant, out, ldo, scratch, st, ev, ch);
    case 5: return launch_dcb<T, MT, 5>(h, src, H, W, quant, out, ldo, scratch, st, ev, ch);
    case 6: return launch_dcb<T, MT, 6>(h, src, H, W, quant, out, ldo, scratch, st, ev, ch);
    case 8: return launch_dcb<T, MT, 8>(h, src, H, W, quant, out, ldo, scratch, st, ev, ch);
    default: dcvc::set_error("DepthConvBlock width %d not instantiated", h->c_p); return dcvc::E_ARG;
    }
}

template <typename T, int MT, int NTW>
int launch_conv(const dcvc_conv* h, const ConvParams& cp, hipStream_t st)
{
    const int grid = ((cp.Ho + Tile<MT>::TH - 1) / Tile<MT>::TH) * ((cp.Wo + Tile<MT>::TW - 1) / Tile<MT>::TW);
    const size_t row = (size_t)(cp.src.c0 + cp.src.c1 + Traits<T>::kPad) * sizeof(T);
    size_t lds = (size_t)Tile<MT>::M * row;
    ConvParams p = cp;
    if (cp.KH * cp.KW > 1) {   // whole input tile + halo in LDS if it fits next to a second workgroup
        const size_t halo = (size_t)((Tile<MT>::TH - 1) * cp.stride + cp.KH) * ((Tile<MT>::TW - 1) * cp.stride + cp.KW) * row;
        // (stride 2 keeps the tap-at-a-time path: rows two pixels apart collide in the LDS banks - measured 37 -> 52 us
        //  for the 3x3 s2 256 -> 128 conv with the halo tile)
        p.halo = cp.stride == 1 && halo <= 96 * 1024;
        if (p.halo) lds = halo;
    }
    // maps that do not fill the GPU with one workgroup per pixel tile: one workgroup per (tile, channel pass)
    const int passes = (cp.N / 16 + NWAVE * NTW - 1) / (NWAVE * NTW);
    p.split_n = passes > 1 && grid < 400;
    const dim3 g(grid, p.split_n ? passes : 1);
    if (p.halo) {
        int rc = set_lds(conv_kernel<T, MT, NTW, true>, lds);
        if (rc) return rc;
        hipLaunchKernelGGL((conv_kernel<T, MT, NTW, true>), g, dim3(NTHREADS), lds, st, p);
    } else {
        int rc = set_lds(conv_kernel<T, MT, NTW, false>, lds);
        if (rc) return rc;
        hipLaunchKernelGGL((conv_kernel<T, MT, NTW, false>), g, dim3(NTHREADS), lds, st, p);
    }
    DCVC_LAUNCH_CHECK();
    (void)h;
    return 0;
}

template <typename T, int MT>
int dispatch_conv(const dcvc_conv* h, const ConvParams& cp, hipStream_t st)
{
    const int nt = cp.N / 16;
    if (nt <= 8) return launch_conv<T, MT, 2>(h, cp, st);
    if (nt <= 12) return launch_conv<T, MT, 3>(h, cp, st);
    return launch_conv<T, MT, 4>(h, cp, st);
}


// Pixel-tile selection (f16): 16*MT pixels per workgroup.  Small feature maps take 32-pixel tiles so
// that the grid still covers the 256 CUs (measured: 68x120 maps 15-25 % faster); 128-pixel tiles
// (MT = 8, one workgroup per CU) were measured slower than 64-pixel tiles at two workgroups per CU
// (80 vs 63 us at C = 256, 136x240) and are not instantiated.  (Developer build: DCVC_MT overrides.)
static int pick_mt_f16(int H, int W, int c_p)
{
#ifdef DCVC_DIAG
    static const int forced = getenv("DCVC_MT") ? atoi(getenv("DCVC_MT")) : 0;
    if (forced == 2 || forced == 4) return forced;
#endif
    const long P = (long)H * W;
    (void)c_p;
    return P >= 12000 ? 4 : 2;
}

template <int MT>
int dispatch_dcb_f16(const dcvc_dcb* h, const SrcPair& src, int H, int W, const float* quant, void* out, int64_t ldo,
                     void* scratch, hipStream_t st, hipEvent_t* ev, ChainArgs ch)
{
    return dispatch_dcb<half_t, MT>(h, src, H, W, quant, out, ldo, scratch, st, ev, ch);
}

static int run_dcb(const dcvc_dcb* h, const SrcPair& src, int H, int W, const float* quant, void* out, int64_t ldo,
                   void* scratch, hipStream_t st, hipEvent_t* ev, ChainArgs ch = ChainArgs())
{
    if (h->dtype != DCVC_F16) return dispatch_dcb<float, 2>(h, src, H, W, quant, out, ldo, scratch, st, ev, ch);
    switch (pick_mt_f16(H, W, h->c_p)) {
    case 2: return dispatch_dcb_f16<2>(h, src, H, W, quant, out, ldo, scratch, st, ev, ch);
    default: return dispatch_dcb_f16<4>(h, src, H, W, quant, out, ldo, scratch, st, ev, ch);
    }
}

static bool c128_enabled()   // DCVC_C128=0: 3x3 convs by conv_kernel (A/B measurements, bit-identity checks)
{
    static const bool on = !(getenv("DCVC_C128") && atoi(getenv("DCVC_C128")) == 0);
    return on;
}

static int run_conv(const dcvc_conv* h, const ConvParams& cp, hipStream_t st)
{
    if (h->dtype != DCVC_F16) return dispatch_conv<float, 2>(h, cp, st);
    if (h->w_s2.p != nullptr && c128_enabled() && cp.src.c1 == 0 && (long)cp.Ho * cp.Wo < 12000) {     // (a workgroup streams all the weights for 32 pixels: small maps)
        t128::ConvS2Params p{};
        p.x = cp.src.x0;
        p.ldx = cp.src.ld0;
        p.H = cp.H;
        p.W = cp.W;
        p.Ho = cp.Ho;
        p.Wo = cp.Wo;
        p.kin = cp.src.c0;
        p.k = cp.KH;
        p.pad = cp.pad;
        p.wt = h->w_s2.p;
        p.b = cp.b;
        p.in_q = cp.in_q;
        p.in_qn = cp.in_qn;
        p.out = cp.out;
        p.ldo = cp.ldo;
        const int ntw = cp.N / 128;
        const size_t lds = t128::conv_s2_lds(p.kin, p.k, ntw);
        const dim3 g(((cp.Ho + t128::S2_TH - 1) / t128::S2_TH) * ((cp.Wo + t128::S2_TW - 1) / t128::S2_TW));
        int rc = ntw == 2 ? set_lds(t128::conv_s2_t32_kernel<2>, lds) : set_lds(t128::conv_s2_t32_kernel<1>, lds);
        if (rc) return rc;
        if (ntw == 2)
            hipLaunchKernelGGL(t128::conv_s2_t32_kernel<2>, g, dim3(256), lds, st, p);
        else
            hipLaunchKernelGGL(t128::conv_s2_t32_kernel<1>, g, dim3(256), lds, st, p);
        DCVC_LAUNCH_CHECK();
        return 0;
    }
    if (h->w_c128[0].p != nullptr && c128_enabled() && cp.src.c1 == 0 && cp.in_q == nullptr) {
        const int tiles = ((cp.H + t128::TH - 1) / t128::TH) * ((cp.W + t128::TW - 1) / t128::TW);
        // 256-channel slices at one workgroup per CU if that grid runs in one round, else 128-channel slices at two per CU
        const int ntw = tiles * (cp.N / 256) <= 256 ? 2 : 1;
        t128::Conv128Params p{};
        p.x = cp.src.x0;
        p.ldx = cp.src.ld0;
        p.H = cp.H;
        p.W = cp.W;
        p.kin = cp.src.c0;
        p.wt = h->w_c128[ntw - 1].p;
        p.b = cp.b;
        p.out = cp.out;
        p.ldo = cp.ldo;
        p.shuffle = cp.epi == DCVC_EPI_SHUFFLE2;
        p.cs_p = cp.cs_p;
        const size_t lds = t128::conv128_lds(p.kin, ntw);
        const dim3 g(tiles, cp.N / (128 * ntw));
        int rc = ntw == 2 ? set_lds(t128::conv3x3_t128_kernel<2>, lds) : set_lds(t128::conv3x3_t128_kernel<1>, lds);
        if (rc) return rc;
        if (ntw == 2)
            hipLaunchKernelGGL(t128::conv3x3_t128_kernel<2>, g, dim3(t128::NTHR), lds, st, p);
        else
            hipLaunchKernelGGL(t128::conv3x3_t128_kernel<1>, g, dim3(t128::NTHR), lds, st, p);
        DCVC_LAUNCH_CHECK();
        return 0;
    }
    const long P = (long)cp.Ho * cp.Wo;
#ifdef DCVC_DIAG
    static const int forced = getenv("DCVC_CONV_MT") ? atoi(getenv("DCVC_CONV_MT")) : 0;
    if (forced == 2) return dispatch_conv<half_t, 2>(h, cp, st);
    if (forced == 4) return dispatch_conv<half_t, 4>(h, cp, st);
    if (forced == 8) return dispatch_conv<half_t, 8>(h, cp, st);
#endif
    if (P >= 12000) return dispatch_conv<half_t, 4>(h, cp, st);
    return dispatch_conv<half_t, 2>(h, cp, st);
}

}  // namespace

extern "C" {

int dcvc_dcb_create(int dtype, int cin, int c, int shortcut, const float* adaptor_w, const float* adaptor_b,
                    const float* w1, const float* b1, const float* wd, const float* bd, const float* w2,
                    const float* b2, const float* w3, const float* b3, const float* w4, const float* b4,
                    dcvc_dcb** out)
{
    DCVC_REQUIRE(out && w1 && b1 && wd && bd && w2 && b2 && w3 && b3 && w4 && b4, "dcvc_dcb_create: null weight pointer");
    DCVC_REQUIRE(dtype == DCVC_F16 || dtype == DCVC_F32, "dcvc_dcb_create: bad dtype %d", dtype);
    DCVC_REQUIRE(c > 0 && cin > 0 && (adaptor_w != nullptr || cin == c), "dcvc_dcb_create: cin %d != c %d needs an adaptor", cin, c);
    std::unique_ptr<dcvc_dcb> h(new dcvc_dcb());
    h->dtype = dtype;
    h->cin = cin;
    h->c = c;
    h->shortcut = shortcut ? 1 : 0;
    h->adapt = adaptor_w != nullptr;
    const int Cp = h->c_p = round_up(c, 64);
    const int Kp = h->cin_p = h->adapt ? round_up(cin, 32) : Cp;
    const int C = c;
    int rc = 0;
    if (h->adapt) {
        rc |= pack_any(dtype, h->wa, Cp, Kp, [&](int n, int k) { return (n < C && k < cin) ? adaptor_w[(size_t)n * cin + k] : 0.f; });
        rc |= upload_f32(h->ba, Cp, [&](int n) { return n < C ? adaptor_b[n] : 0.f; });
    }
    // fp16 mode evaluates both activations on pre-scaled pre-activations (Traits<half_t>::gate / gate2):
    // dc.0 and ffn.0 are packed multiplied by kAct, their consumers (depthwise taps, ffn.2) divided by
    // it; exact mode packs everything unchanged (ka == 1)
    const float ka = dtype == DCVC_F16 ? Traits<half_t>::kAct : 1.0f;
    rc |= pack_any(dtype, h->w1, Cp, Cp, [&](int n, int k) { return (n < C && k < C) ? ka * w1[(size_t)n * C + k] : 0.f; });
    rc |= upload_f32(h->b1, Cp, [&](int n) { return n < C ? ka * b1[n] : 0.f; });
    auto dwget = [&](int i) { const int t = i / Cp, ch = i % Cp; return ch < C ? wd[(size_t)ch * 9 + t] / ka : 0.f; };
    rc |= dtype == DCVC_F16 ? upload_T<half_t>(h->wd, 9 * Cp, dwget) : upload_T<float>(h->wd, 9 * Cp, dwget);
    rc |= upload_f32(h->bd, Cp, [&](int n) { return n < C ? bd[n] : 0.f; });
    rc |= pack_any(dtype, h->w2, Cp, Cp, [&](int n, int k) { return (n < C && k < C) ? w2[(size_t)n * C + k] : 0.f; });
    rc |= upload_f32(h->b2, Cp, [&](int n) { return n < C ? b2[n] : 0.f; });
    // ffn.0: logical rows [0,2C) pair with rows [2C,4C) (WSiLUChunkAdd); physical halves are 2*Cp wide
    auto u_row = [&](int n) { const int half = n / (2 * Cp), cc = n % (2 * Cp); return cc < 2 * C ? half * 2 * C + cc : -1; };
    rc |= pack_any(dtype, h->w3, 4 * Cp, Cp, [&](int n, int k) { const int rr = u_row(n); return (rr >= 0 && k < C) ? ka * w3[(size_t)rr * C + k] : 0.f; });
    rc |= upload_f32(h->b3, 4 * Cp, [&](int n) { const int rr = u_row(n); return rr >= 0 ? ka * b3[rr] : 0.f; });
    rc |= pack_any(dtype, h->w4, Cp, 2 * Cp, [&](int n, int k) { return (n < C && k < 2 * C) ? w4[(size_t)n * 2 * C + k] / ka : 0.f; });
    rc |= upload_f32(h->b4, Cp, [&](int n) { return n < C ? b4[n] : 0.f; });
    if (rc == 0 && dtype == DCVC_F16 && t128_width_ok(Cp)) {
        // the same (pre-scaled, fp16) tail weights once more, as the fragment streams of dcb_tail128_kernel
        auto W2 = [&](int n, int k) { return (n < C && k < C) ? w2[(size_t)n * C + k] : 0.f; };
        auto W3 = [&](int n, int k) { const int rr = u_row(n); return (rr >= 0 && k < C) ? ka * w3[(size_t)rr * C + k] : 0.f; };
        auto W4 = [&](int n, int k) { return (n < C && k < 2 * C) ? w4[(size_t)n * 2 * C + k] / ka : 0.f; };
        if (h->adapt && Kp % 64 == 0) {
            auto WA = [&](int n, int k) { return (n < C && k < cin) ? adaptor_w[(size_t)n * cin + k] : 0.f; };
            rc |= Cp == 128 ? pack_t128_rect<128>(h->wa_t128, Kp, WA) : Cp == 256 ? pack_t128_rect<256>(h->wa_t128, Kp, WA) : Cp == 320 ? pack_t128_rect<320>(h->wa_t128, Kp, WA)
                            : Cp == 384 ? pack_t128_rect<384>(h->wa_t128, Kp, WA) : pack_t128_rect<512>(h->wa_t128, Kp, WA);
        }
        rc |= pack_t128_square_any(Cp, h->w1_t128, [&](int n, int k) { return (n < C && k < C) ? ka * w1[(size_t)n * C + k] : 0.f; });
        rc |= Cp == 128 ? pack_t128<128>(h->wt128, W2, W3, W4) : Cp == 256 ? pack_t128<256>(h->wt128, W2, W3, W4) : Cp == 320 ? pack_t128<320>(h->wt128, W2, W3, W4)
                        : Cp == 384 ? pack_t128<384>(h->wt128, W2, W3, W4) : pack_t128<512>(h->wt128, W2, W3, W4);
    }
    if (rc) return rc < 0 ? rc : dcvc::E_MEM;
    *out = h.release();
    return 0;
}

void dcvc_dcb_destroy(dcvc_dcb* h) { delete h; }

size_t dcvc_dcb_scratch_bytes(const dcvc_dcb* h, int H, int W)
{
    if (!h) return 0;
    return (size_t)H * W * h->c_p * dcvc::elem_size(h->dtype) * 3;   // two `a` slots (chained blocks alternate) + x'
}

int dcvc_dcb_forward(const dcvc_dcb* h, const void* x0, int64_t ld0, int c0, const void* x1, int64_t ld1, int c1,
                     int H, int W, const float* quant, void* out, int64_t ldo, void* scratch, void* stream)
{
    return dcvc_dcb_forward_chained(h, x0, ld0, c0, x1, ld1, c1, H, W, quant, out, ldo, scratch, stream, 0, 0, nullptr);
}

int dcvc_dcb_forward_chained(const dcvc_dcb* h, const void* x0, int64_t ld0, int c0, const void* x1, int64_t ld1, int c1,
                             int H, int W, const float* quant, void* out, int64_t ldo, void* scratch, void* stream,
                             int head_done, int a_slot, const dcvc_dcb* next)
{
    DCVC_REQUIRE(h && x0 && out && scratch, "dcvc_dcb_forward: null pointer");
    DCVC_REQUIRE(!head_done || !h->adapt, "dcvc_dcb_forward_chained: a block with adaptor computes its own head");
    if (next)
        DCVC_REQUIRE(!next->adapt && next->c_p == h->c_p && next->dtype == h->dtype && !quant && !h->shortcut,
                     "dcvc_dcb_forward_chained: cannot fuse the next block's head (needs same width and type, no "
                     "adaptor there, no shortcut / quant step here)");
    DCVC_REQUIRE(H > 0 && W > 0, "dcvc_dcb_forward: empty input %dx%d", H, W);
    DCVC_REQUIRE(c0 % 32 == 0 && c1 % 32 == 0 && c0 + c1 == h->cin_p,
                 "dcvc_dcb_forward: input channels %d+%d do not match the block (%d physical)", c0, c1, h->cin_p);
    DCVC_REQUIRE((x1 != nullptr) == (c1 > 0), "dcvc_dcb_forward: x1/c1 mismatch");
    DCVC_REQUIRE(h->adapt || c1 == 0, "dcvc_dcb_forward: a block without adaptor takes a single source");
    DCVC_REQUIRE(ld0 >= c0 && ldo >= h->c_p && (c1 == 0 || ld1 >= c1), "dcvc_dcb_forward: row stride too small");
    const size_t es = dcvc::elem_size(h->dtype);
    DCVC_REQUIRE(((uintptr_t)x0 % 16) == 0 && (ld0 * es) % 16 == 0 && (c1 == 0 || (((uintptr_t)x1 % 16) == 0 && (ld1 * es) % 16 == 0)),
                 "dcvc_dcb_forward: inputs must be 16-byte aligned");
    SrcPair src{x0, (long)ld0, c0, x1, (long)ld1, c1};
    hipStream_t st = (hipStream_t)stream;
    ChainArgs ch;
    ch.head_done = head_done;
    ch.a_slot = a_slot;
    ch.next = next;
    return run_dcb(h, src, H, W, quant, out, ldo, scratch, st, nullptr, ch);
}

int dcvc_dcb_forward_then_conv(const dcvc_dcb* h, const void* x0, int64_t ld0, int c0, const void* x1, int64_t ld1, int c1,
                               int H, int W, void* scratch, void* stream, int head_done, int a_slot,
                               const dcvc_conv* conv, const float* conv_quant, void* conv_out, int64_t ldco)
{
    DCVC_REQUIRE(h && x0 && scratch && conv && conv_out, "dcvc_dcb_forward_then_conv: null pointer");
    DCVC_REQUIRE(!head_done || !h->adapt, "dcvc_dcb_forward_then_conv: a block with adaptor computes its own head");
    DCVC_REQUIRE(!h->shortcut, "dcvc_dcb_forward_then_conv: a block with shortcut cannot feed a fused conv");
    DCVC_REQUIRE(conv->dtype == h->dtype && conv->kh == 1 && conv->kw == 1 && conv->stride == 1 && conv->pad == 0 &&
                     conv->cin_p == h->c_p && conv->n_p == h->c_p && conv->cin == h->c &&
                     (conv->epi == DCVC_EPI_BIAS || conv->epi == DCVC_EPI_BIAS_QUANT),
                 "dcvc_dcb_forward_then_conv: the conv must be 1x1, stride 1, of the block's width (%d -> %d given, block %d), "
                 "epilogue bias or bias*quant", conv->cin, conv->cout, h->c);
    DCVC_REQUIRE((conv->epi == DCVC_EPI_BIAS_QUANT) == (conv_quant != nullptr), "dcvc_dcb_forward_then_conv: quant vector / epilogue mismatch");
    DCVC_REQUIRE(H > 0 && W > 0, "dcvc_dcb_forward_then_conv: empty input %dx%d", H, W);
    DCVC_REQUIRE(c0 % 32 == 0 && c1 % 32 == 0 && c0 + c1 == h->cin_p,
                 "dcvc_dcb_forward_then_conv: input channels %d+%d do not match the block (%d physical)", c0, c1, h->cin_p);
    DCVC_REQUIRE((x1 != nullptr) == (c1 > 0), "dcvc_dcb_forward_then_conv: x1/c1 mismatch");
    DCVC_REQUIRE(h->adapt || c1 == 0, "dcvc_dcb_forward_then_conv: a block without adaptor takes a single source");
    DCVC_REQUIRE(ld0 >= c0 && ldco >= h->c_p && (c1 == 0 || ld1 >= c1), "dcvc_dcb_forward_then_conv: row stride too small");
    const size_t es = dcvc::elem_size(h->dtype);
    DCVC_REQUIRE(((uintptr_t)x0 % 16) == 0 && (ld0 * es) % 16 == 0 && ((uintptr_t)conv_out % 16) == 0 && (ldco * es) % 16 == 0 &&
                     (c1 == 0 || (((uintptr_t)x1 % 16) == 0 && (ld1 * es) % 16 == 0)),
                 "dcvc_dcb_forward_then_conv: buffers must be 16-byte aligned");
    SrcPair src{x0, (long)ld0, c0, x1, (long)ld1, c1};
    ChainArgs ch;
    ch.head_done = head_done;
    ch.a_slot = a_slot;
    ch.conv = conv;
    ch.conv_q = conv_quant;
    ch.conv_out = conv_out;
    ch.ldco = ldco;
    return run_dcb(h, src, H, W, nullptr, nullptr, 0, scratch, (hipStream_t)stream, nullptr, ch);
}

int dcvc_dcb_profile(const dcvc_dcb* h, const void* x0, int64_t ld0, int c0, int H, int W, void* out, int64_t ldo,
                     void* scratch, void* stream, int iters, float* head_ms, float* tail_ms)
{
    DCVC_REQUIRE(h && x0 && out && scratch && head_ms && tail_ms && iters > 0, "dcvc_dcb_profile: bad arguments");
    DCVC_REQUIRE(c0 == h->cin_p && ld0 >= c0 && ldo >= h->c_p, "dcvc_dcb_profile: shape mismatch");
    SrcPair src{x0, (long)ld0, c0, nullptr, 0, 0};
    hipStream_t st = (hipStream_t)stream;
    std::vector<hipEvent_t> ev((size_t)iters * 3);
    for (auto& e : ev) DCVC_HIP(hipEventCreate(&e));
    int rc = 0;
    for (int i = 0; i < iters && rc == 0; ++i)
        rc = run_dcb(h, src, H, W, nullptr, out, ldo, scratch, st, &ev[3 * i]);
    if (rc == 0) {
        DCVC_HIP(hipStreamSynchronize(st));
        double th = 0, tt = 0;
        for (int i = 0; i < iters; ++i) {
            float a = 0, b = 0;
            DCVC_HIP(hipEventElapsedTime(&a, ev[3 * i], ev[3 * i + 1]));
            DCVC_HIP(hipEventElapsedTime(&b, ev[3 * i + 1], ev[3 * i + 2]));
            th += a;
            tt += b;
        }
        *head_ms = (float)(th / iters);
        *tail_ms = (float)(tt / iters);
    }
    for (auto& e : ev) (void)hipEventDestroy(e);
    return rc;
}

int dcvc_dcb_profile_tail(const dcvc_dcb* h, const void* x0, int64_t ld0, int c0, int H, int W, void* out, int64_t ldo,
                          void* scratch, void* stream, int iters, float* tail_ms)
{
    DCVC_REQUIRE(h && x0 && out && scratch && tail_ms && iters > 0, "dcvc_dcb_profile_tail: bad arguments");
    DCVC_REQUIRE(c0 == h->cin_p && ld0 >= c0 && ldo >= h->c_p && !h->adapt, "dcvc_dcb_profile_tail: shape mismatch or block with adaptor");
    SrcPair src{x0, (long)ld0, c0, nullptr, 0, 0};
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t e0, e1;
    DCVC_HIP(hipEventCreate(&e0));
    DCVC_HIP(hipEventCreate(&e1));
    ChainArgs first;
    first.separate_head = 1;
    int rc = run_dcb(h, src, H, W, nullptr, out, ldo, scratch, st, nullptr, first);   // head + tail once: `a` is in the scratch
    ChainArgs tail_only;
    tail_only.head_done = 1;
    if (rc == 0) rc = hipEventRecord(e0, st) == hipSuccess ? 0 : dcvc::E_HIP;
    for (int i = 0; i < iters && rc == 0; ++i) rc = run_dcb(h, src, H, W, nullptr, out, ldo, scratch, st, nullptr, tail_only);
    if (rc == 0) {
        float ms = 0;
        DCVC_HIP(hipEventRecord(e1, st));
        DCVC_HIP(hipStreamSynchronize(st));
        DCVC_HIP(hipEventElapsedTime(&ms, e0, e1));
        *tail_ms = ms / iters;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

int dcvc_conv_create(int dtype, int cin, int cout, int kh, int kw, int stride, int pad, int epilogue,
                     const float* w, const float* b, dcvc_conv** out)
{
    DCVC_REQUIRE(out && w && b, "dcvc_conv_create: null pointer");
    DCVC_REQUIRE(dtype == DCVC_F16 || dtype == DCVC_F32, "dcvc_conv_create: bad dtype %d", dtype);
    DCVC_REQUIRE(kh >= 1 && kh <= 3 && kw >= 1 && kw <= 3 && stride >= 1 && stride <= 2 && pad >= 0 && pad <= 1,
                 "dcvc_conv_create: unsupported geometry k=%dx%d s=%d p=%d", kh, kw, stride, pad);
    DCVC_REQUIRE(epilogue >= DCVC_EPI_BIAS && epilogue <= DCVC_EPI_WSILU, "dcvc_conv_create: bad epilogue %d", epilogue);
    DCVC_REQUIRE(epilogue != DCVC_EPI_SHUFFLE2 || cout % 4 == 0, "dcvc_conv_create: shuffle needs cout %% 4 == 0");
    std::unique_ptr<dcvc_conv> h(new dcvc_conv());
    h->dtype = dtype;
    h->cin = cin;
    h->cout = cout;
    h->kh = kh;
    h->kw = kw;
    h->stride = stride;
    h->pad = pad;
    h->epi = epilogue;
    const int Kp = h->cin_p = round_up(cin, 32);
    const int taps = kh * kw;
    int Np;
    std::function<int(int)> to_log;   // physical output channel -> logical (or -1)
    if (epilogue == DCVC_EPI_SHUFFLE2) {
        const int cs = cout / 4, csp = h->cs_p = round_up(cs, 32);
        Np = 4 * csp;
        to_log = [=](int n) { const int s = n / csp, cc = n % csp; return cc < cs ? cc * 4 + s : -1; };
    } else {
        h->cs_p = 0;
        Np = round_up(cout, 32);
        to_log = [=](int n) { return n < cout ? n : -1; };
    }
    h->n_p = Np;
    int rc = pack_any(dtype, h->w, Np, taps * Kp, [&](int n, int k) {
        const int nl = to_log(n), t = k / Kp, ci = k % Kp;
        return (nl >= 0 && ci < cin) ? w[((size_t)nl * cin + ci) * taps + t] : 0.f;
    });
    rc |= upload_f32(h->b, Np, [&](int n) { const int nl = to_log(n); return nl >= 0 ? b[nl] : 0.f; });
    if (rc == 0 && dtype == DCVC_F16 && taps == 1 && epilogue != DCVC_EPI_SHUFFLE2 && Np == Kp && Np == round_up(cout, 64) && t128_width_ok(Np))
        rc |= pack_t128_square_any(Np, h->w_t128, [&](int n, int k) { return (n < cout && k < cin) ? w[(size_t)n * cin + k] : 0.f; });
    if (rc == 0 && dtype == DCVC_F16 && kh == 3 && kw == 3 && stride == 1 && pad == 1 && Kp % 64 == 0 && Kp <= 256 && Np % 256 == 0 &&
        (epilogue == DCVC_EPI_BIAS || (epilogue == DCVC_EPI_SHUFFLE2 && h->cs_p % 256 == 0))) {
        auto WC = [&](int n, int t, int ci) {
            const int nl = to_log(n);
            return (nl >= 0 && ci < cin) ? w[((size_t)nl * cin + ci) * taps + t] : 0.f;
        };
        rc |= pack_t128_conv(h->w_c128[0], 1, 9, 4, Np, Kp, WC);
        rc |= pack_t128_conv(h->w_c128[1], 2, 9, 8, Np, Kp, WC);
    }
    if (rc == 0 && dtype == DCVC_F16 && stride == 2 && ((kh == 2 && kw == 2 && pad == 0) || (kh == 3 && kw == 3 && pad == 1)) &&
        Kp % 128 == 0 && Kp <= 384 && (Np == 128 || Np == 256) && epilogue == DCVC_EPI_BIAS)
        rc |= pack_t128_conv(h->w_s2, Np / 128, taps, 8 * (Np / 128), Np, Kp, [&](int n, int t, int ci) {
            const int nl = to_log(n);
            return (nl >= 0 && ci < cin) ? w[((size_t)nl * cin + ci) * taps + t] : 0.f;
        });
    if (rc) return rc < 0 ? rc : dcvc::E_MEM;
    *out = h.release();
    return 0;
}

void dcvc_conv_destroy(dcvc_conv* h) { delete h; }

int dcvc_conv_forward(const dcvc_conv* h, const void* x0, int64_t ld0, int c0, const void* x1, int64_t ld1, int c1,
                      int H, int W, const float* quant, void* out, int64_t ldo, void* stream)
{
    return dcvc_conv_forward_scaled(h, x0, ld0, c0, x1, ld1, c1, H, W, nullptr, quant, out, ldo, stream);
}

int dcvc_conv_forward_scaled(const dcvc_conv* h, const void* x0, int64_t ld0, int c0, const void* x1, int64_t ld1, int c1,
                             int H, int W, const float* in_scale, const float* quant, void* out, int64_t ldo, void* stream)
{
    DCVC_REQUIRE(h && x0 && out, "dcvc_conv_forward: null pointer");
    DCVC_REQUIRE(H > 0 && W > 0, "dcvc_conv_forward: empty input %dx%d", H, W);
    DCVC_REQUIRE(c0 % 32 == 0 && c1 % 32 == 0 && c0 + c1 == h->cin_p,
                 "dcvc_conv_forward: input channels %d+%d do not match the layer (%d physical)", c0, c1, h->cin_p);
    DCVC_REQUIRE((x1 != nullptr) == (c1 > 0), "dcvc_conv_forward: x1/c1 mismatch");
    DCVC_REQUIRE(h->epi != DCVC_EPI_BIAS_QUANT || quant != nullptr, "dcvc_conv_forward: quant vector required");
    const size_t es = dcvc::elem_size(h->dtype);
    DCVC_REQUIRE(((uintptr_t)x0 % 16) == 0 && (ld0 * es) % 16 == 0 && (c1 == 0 || (((uintptr_t)x1 % 16) == 0 && (ld1 * es) % 16 == 0)),
                 "dcvc_conv_forward: inputs must be 16-byte aligned");
    ConvParams cp{};
    cp.src = SrcPair{x0, (long)ld0, c0, x1, (long)ld1, c1};
    cp.H = H;
    cp.W = W;
    cp.Ho = (H + 2 * h->pad - h->kh) / h->stride + 1;
    cp.Wo = (W + 2 * h->pad - h->kw) / h->stride + 1;
    DCVC_REQUIRE(cp.Ho > 0 && cp.Wo > 0, "dcvc_conv_forward: empty output");
    cp.KH = h->kh;
    cp.KW = h->kw;
    cp.stride = h->stride;
    cp.pad = h->pad;
    cp.N = h->n_p;
    cp.n_log = h->cout;
    cp.cs_p = h->cs_p;
    cp.w = h->w.p;
    cp.b = (const float*)h->b.p;
    cp.q = quant;
    cp.in_q = in_scale;
    cp.in_qn = h->cin;
    cp.epi = h->epi;
    cp.out = out;
    cp.ldo = ldo;
    DCVC_REQUIRE(ldo >= (h->epi == DCVC_EPI_SHUFFLE2 ? h->cs_p : h->n_p), "dcvc_conv_forward: output row stride too small");
    hipStream_t st = (hipStream_t)stream;
    return run_conv(h, cp, st);
}

}  // extern "C"
