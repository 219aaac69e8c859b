// gemm_core.hpp - MFMA building blocks shared by the DCVC-RT conv kernels (gfx950 / CDNA4).
//
// Work decomposition used by every kernel in dcvc_nn.hip:
//   * a workgroup (256 threads = 4 waves, one per SIMD) owns M = 16*MT output pixels and ALL
//     output channels of the layer, so chains of 1x1 convolutions stay on chip (LDS) between GEMMs;
//   * the activation tile (A operand) lives in LDS as [M][K + pad]; every wave reads all of it;
//   * the output channels are split over the 4 waves in 16-wide n-tiles (tile t -> wave t % 4),
//     so every weight element is needed by exactly ONE wave: weights are pre-packed on the host
//     into MFMA-fragment order and streamed global/L2 -> VGPR with one coalesced 16-byte (f16) or
//     32-byte (f32) load per lane, never touching LDS and needing no barrier inside a K loop;
//   * the weights are the MFMA "A" operand and the pixels the "B" operand, so a lane ends up with
//     4 consecutive channels of one pixel (packed 8-byte epilogue accesses);
//   * fp16 path: v_mfma_f32_16x16x32_f16 (fp32 accumulate);  fp32 "exact" path:
//     v_mfma_f32_16x16x4_f32, whose result is bit-for-bit a k-ascending fmaf chain
//     (MI355X_MICROARCH.md, Matrix cores), which oracle/nn_oracle.c reproduces on the CPU.
//
// Reduction dimension is processed in groups of KG = 32 elements.  Lane l (r = l & 15, q = l >> 4)
// consumes 8 consecutive STORED elements [32*g + 8*q, +8) of row r of both operands per group:
//   f16: stored order = natural order (one 16x16x32 MFMA per group);
//   f32: element k = 4*j + q of the group is stored at q*8 + j (perm32), so the same 8
//        consecutive stored elements feed eight 16x16x4 MFMAs j = 0..7 in ascending-k order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dcvc_math.h"

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx8 __attribute__((ext_vector_type(8)));

constexpr int KG = 32;
constexpr int NWAVE = 4;
constexpr int NTHREADS = 256;

template <typename T>
struct Traits;

template <>
struct Traits<half_t> {
    using frag_t = half8;
    static constexpr int kPad = 16;   // LDS row padding, elements (32 bytes)
    static constexpr int kVec = 8;    // elements per 16-byte vector
    static __host__ __device__ __forceinline__ int perm(int k) { return k; }
    static __device__ __forceinline__ floatx4 mma(const frag_t& a, const frag_t& b, floatx4 c)
    {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ float to_f(half_t v) { return (float)v; }
    // fp32 -> fp16 (RNE) of the fp32 VALUE.  The empty asm hides the value's producer: otherwise the compiler folds a
    // preceding fp32 multiply into v_fma_mixlo_f16 (product rounded ONCE, straight to fp16) for whichever elements its
    // scheduler picks, and v_pk_mul_f32 + v_cvt_pk_f16_f32 (rounded twice) for the rest - a choice that differs from
    // kernel to kernel, so the same gate computed in dcb_head_kernel and in a fused tail disagreed by one fp16 ulp on
    // values within 2 fp32 ulps of a rounding midpoint (2 of 4 M).  With it every store is the two-step rounding the
    // oracle models, whatever kernel it sits in.
    static __device__ __forceinline__ half_t from_f(float v)
    {
        asm("" : "+v"(v));   // free: dcb_tail_kernel 51.5 us with it, 54.6 us without on the same box (tools/kbench.py)
        return (half_t)v;
    }
    // fp16 mode: hardware exp2 / rcp (about 1 ulp in fp32, far below the fp16 storage rounding);
    // 5 VALU instructions instead of ~35 for the reproducible dcvc_wsiluf.
    static __device__ __forceinline__ float wsilu(float x)
    {
        const float e = __builtin_amdgcn_exp2f(x * -5.770780163555854f);   // exp(-4x)
        return x * __builtin_amdgcn_rcpf(1.0f + e);
    }
    // Pre-scaled activations.  wsilu(u) = u / (1 + 2^(kAct u)) with kAct = -4 log2(e).  When the layer
    // that produces u has its weights and bias multiplied by kAct at pack time (u' = kAct u) and the
    // layer that consumes the result has its weights divided by kAct, the kernel only evaluates
    //   g(u') = u' / (1 + 2^u') = kAct wsilu(u)
    // - 4 VALU instructions per activation, and the bias can seed the accumulator (no separate add).
    static constexpr bool kActPrescaled = true;
    static constexpr float kAct = -5.770780163555854f;
    static __device__ __forceinline__ float gate(float up) { return up * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(up)); }
    static __device__ __forceinline__ float gate2(float lo, float hi)   // g(lo) + g(hi)
    {
        const float a = lo * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(lo));
        return DCVC_FMAF(hi, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(hi)), a);
    }
};

template <>
struct Traits<float> {
    using frag_t = floatx8;
    static constexpr int kPad = 8;
    static constexpr int kVec = 4;
    static __host__ __device__ __forceinline__ int perm(int k)
    {
        return (k & ~31) | ((k & 3) << 3) | ((k & 31) >> 2);
    }
    static __device__ __forceinline__ floatx4 mma(const frag_t& a, const frag_t& b, floatx4 c)
    {
#pragma unroll
        for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], c, 0, 0, 0);
        return c;
    }
    static __device__ __forceinline__ float to_f(float v) { return v; }
    static __device__ __forceinline__ float from_f(float v) { return v; }
    static __device__ __forceinline__ float wsilu(float x) { return dcvc_wsiluf(x); }   // bit-reproducible
    static constexpr bool kActPrescaled = false;   // exact mode keeps the reference's operation order
    static constexpr float kAct = 1.0f;
    static __device__ __forceinline__ float gate(float u) { return dcvc_wsiluf(u); }
    static __device__ __forceinline__ float gate2(float lo, float hi) { return dcvc_wsiluf(lo) + dcvc_wsiluf(hi); }
};

// acc[m][i] += W[n-tile tiles[i]][K] * X[pixel-tile m][K]^T over `kgs` reduction groups, with the
// WEIGHTS as the MFMA "A" operand and the activation tile as the "B" operand, so the result tile is
// channel-major in registers: lane l holds, for pixel m*16 + (l & 15), the four consecutive channels
// tiles[i]*16 + 4*(l >> 4) + {0,1,2,3}  ->  8-byte (f16) packed LDS / global accesses in epilogues.
//   X      : LDS, row-major [16*MT][ldx] in stored (perm) order, groups 0..kgs-1
//   Wp     : packed weights, fragment (tile, group g) at Wp[(tile * kgs_total + g) * 64 + lane]
//   kg0    : first group of W to use (X group g pairs with W group kg0 + g)
// Weight fragments go straight from L2 into registers (no LDS, no barrier in the loop), PF groups
// ahead.  The first PF groups can be requested early with gemm_prefetch() - before the barrier /
// epilogue that precedes the GEMM - so the ~1 us L2 latency of a cold start is not exposed
// (one wave per SIMD per workgroup has nobody else to hide it).
template <typename T, int NT, int PF>
struct WPre {
    typename Traits<T>::frag_t w[PF][NT];
};

template <typename T, int NT, int PF>
__device__ __forceinline__ void gemm_prefetch(WPre<T, NT, PF>& pre, int kgs,
                                              const typename Traits<T>::frag_t* __restrict__ Wp, int kgs_total,
                                              int kg0, const int (&tiles)[NT], int lane)
{
#pragma unroll
    for (int s = 0; s < PF; ++s) {
        const int g = s < kgs ? s : kgs - 1;
#pragma unroll
        for (int i = 0; i < NT; ++i) pre.w[s][i] = Wp[((size_t)tiles[i] * kgs_total + kg0 + g) * 64 + lane];
    }
}

template <typename T, int MT, int NT, int PF>
__device__ __forceinline__ void gemm_run(floatx4 (&acc)[MT][NT], const T* X, int ldx, int kgs, WPre<T, NT, PF>& pre,
                                         const typename Traits<T>::frag_t* __restrict__ Wp, int kgs_total, int kg0,
                                         const int (&tiles)[NT], int lane)
{
    using frag_t = typename Traits<T>::frag_t;
    const int r = lane & 15, q = lane >> 4;
    const T* x_base = X + r * ldx + q * 8;
    const frag_t* w_base[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) w_base[i] = Wp + ((size_t)tiles[i] * kgs_total + kg0) * 64 + lane;

    // One k-group: MFMAs of fragment i, then refill its registers right away with group g + PF.
    // The refill is UNCONDITIONAL (index clamped to the last group): a branch around a global load
    // makes hipcc give up counting and emit s_waitcnt vmcnt(0) at the loop head, which waits for the
    // loads issued one step earlier, i.e. exposes the full L2 latency in every step (measured: the
    // GEMM phases ran at a quarter of the MFMA issue rate).
    auto step = [&](int g, int s) __attribute__((always_inline)) {
        frag_t x[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) x[m] = *reinterpret_cast<const frag_t*>(x_base + m * 16 * ldx + g * KG);
        const int gn = (g + PF < kgs) ? g + PF : kgs - 1;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m][i] = Traits<T>::mma(pre.w[s][i], x[m], acc[m][i]);
            pre.w[s][i] = w_base[i][(size_t)gn * 64];
        }
        // pin the interleave: [LDS reads] then per fragment [its MFMAs][its refill].  Left alone, the
        // scheduler sinks all refills to the end of the loop body and the next iteration waits on them.
        // (Double-buffering the pixel fragments as well was measured: no gain at two waves per SIMD.)
        constexpr int kMfmaPerMma = sizeof(T) == 2 ? 1 : 8, kVecPerFrag = sizeof(T) == 2 ? 1 : 2;
        __builtin_amdgcn_sched_group_barrier(0x100, MT * kVecPerFrag, 0);
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, MT * kMfmaPerMma, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, kVecPerFrag, 0);
        }
    };
    const int kfull = (kgs / PF) * PF;
    // The GEMM loops run at raised wave priority: the SIMD's vector issue is arbitrated by priority, then age, so a
    // co-resident wave in an epilogue / depthwise (VALU) phase otherwise starves this wave's MFMAs, which need only 8 of
    // every 16 issue cycles and leave the rest to it.  Measured (tools/kbench.py, same box): C=320 136x240 on 8 waves
    // 95.4 -> 84.0 us, C=256 on 2 x 4 waves 54.0 -> 52.8 us; priority 3 = priority 1.
    __builtin_amdgcn_s_setprio(1);
    for (int g0 = 0; g0 < kfull; g0 += PF) {
#pragma unroll
        for (int s = 0; s < PF; ++s) step(g0 + s, s);
    }
#pragma unroll
    for (int s = 0; s < PF - 1; ++s)        // remainder groups (kgs % PF), outside the hot loop
        if (kfull + s < kgs) step(kfull + s, s);
    __builtin_amdgcn_s_setprio(0);
}

// Implicit-GEMM form for KH x KW convolutions: the reduction runs over (tap, cin) in one uninterrupted
// weight pipeline; the pixel fragment of tap (ky, kx) is read from a halo tile in LDS at a per-tap row offset
// (xrow[m] points at the lane's pixel row for tap (0, 0); rows of the halo tile are HT_W pixels wide).
// Same accumulation order as staging one tap at a time (taps outer, channels ascending).
template <typename T, int MT, int NT, int PF>
__device__ __forceinline__ void gemm_taps(floatx4 (&acc)[MT][NT], const T* const (&xrow)[MT], int ldx, int taps, int KW,
                                          int HT_W, int kgin, const typename Traits<T>::frag_t* __restrict__ Wp,
                                          const int (&tiles)[NT], int lane)
{
    using frag_t = typename Traits<T>::frag_t;
    const int kgs = taps * kgin;
    WPre<T, NT, PF> pre;
    gemm_prefetch<T, NT, PF>(pre, kgs, Wp, kgs, 0, tiles, lane);
    const frag_t* w_base[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) w_base[i] = Wp + (size_t)tiles[i] * kgs * 64 + lane;
    int kk = 0, kx = 0, ky = 0;
    auto step = [&](int g, int s) __attribute__((always_inline)) {
        const int off = (ky * HT_W + kx) * ldx + kk * KG;
        if (++kk == kgin) {
            kk = 0;
            if (++kx == KW) {
                kx = 0;
                ++ky;
            }
        }
        frag_t x[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) x[m] = *reinterpret_cast<const frag_t*>(xrow[m] + off);
        const int gn = (g + PF < kgs) ? g + PF : kgs - 1;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m][i] = Traits<T>::mma(pre.w[s][i], x[m], acc[m][i]);
            pre.w[s][i] = w_base[i][(size_t)gn * 64];
        }
        constexpr int kMfmaPerMma = sizeof(T) == 2 ? 1 : 8, kVecPerFrag = sizeof(T) == 2 ? 1 : 2;
        __builtin_amdgcn_sched_group_barrier(0x100, MT * kVecPerFrag, 0);
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, MT * kMfmaPerMma, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, kVecPerFrag, 0);
        }
    };
    const int kfull = (kgs / PF) * PF;
    __builtin_amdgcn_s_setprio(1);   // as in gemm_run
    for (int g0 = 0; g0 < kfull; g0 += PF) {
#pragma unroll
        for (int s = 0; s < PF; ++s) step(g0 + s, s);
    }
#pragma unroll
    for (int s = 0; s < PF - 1; ++s)
        if (kfull + s < kgs) step(kfull + s, s);
    __builtin_amdgcn_s_setprio(0);
}

template <typename T, int MT, int NT, int PF>
__device__ __forceinline__ void gemm_acc(floatx4 (&acc)[MT][NT], const T* X, int ldx, int kgs,
                                         const typename Traits<T>::frag_t* __restrict__ Wp,
                                         int kgs_total, int kg0, const int (&tiles)[NT], int lane)
{
    WPre<T, NT, PF> pre;
    gemm_prefetch<T, NT, PF>(pre, kgs, Wp, kgs_total, kg0, tiles, lane);
    gemm_run<T, MT, NT, PF>(acc, X, ldx, kgs, pre, Wp, kgs_total, kg0, tiles, lane);
}

template <int MT, int NT>
__device__ __forceinline__ void zero_acc(floatx4 (&acc)[MT][NT])
{
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int i = 0; i < NT; ++i) acc[m][i] = floatx4{0.f, 0.f, 0.f, 0.f};
}

// 16-byte vector load/store helpers ------------------------------------------------------------
typedef uint32_t Vec16 __attribute__((ext_vector_type(4)));   // a raw 16-byte vector (stays in VGPRs)
#define VEC16_ZERO (Vec16{0u, 0u, 0u, 0u})

template <typename T>
__device__ __forceinline__ void unpack16(const Vec16& v, float (&f)[Traits<T>::kVec]);

template <>
__device__ __forceinline__ void unpack16<half_t>(const Vec16& v, float (&f)[8])
{
    const half8 h = __builtin_bit_cast(half8, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (float)h[j];
}
template <>
__device__ __forceinline__ void unpack16<float>(const Vec16& v, float (&f)[4])
{
    const floatx4 h = __builtin_bit_cast(floatx4, v);
#pragma unroll
    for (int j = 0; j < 4; ++j) f[j] = h[j];
}

// s[j] = fma(a[j], w[j], s[j]) over the kVec elements of two raw 16-byte vectors, fp32 accumulate.
// fp16: v_fma_mix_f32 reads the halves straight out of the packed dwords (no conversions; hipcc otherwise
// emits v_cvt_f32_f16 per element + v_pk_fma_f32); fp32: the explicit fmaf chain of the exact mode.
template <typename T>
__device__ __forceinline__ void fma_vec16(const Vec16& a, const Vec16& w, float (&s)[Traits<T>::kVec]);

template <>
__device__ __forceinline__ void fma_vec16<half_t>(const Vec16& a, const Vec16& w, float (&s)[8])
{
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "+v"(s[2 * d]) : "v"(a[d]), "v"(w[d]));
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "+v"(s[2 * d + 1]) : "v"(a[d]), "v"(w[d]));
    }
}
template <>
__device__ __forceinline__ void fma_vec16<float>(const Vec16& a, const Vec16& w, float (&s)[4])
{
    const floatx4 af = __builtin_bit_cast(floatx4, a), wf = __builtin_bit_cast(floatx4, w);
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] = DCVC_FMAF(af[j], wf[j], s[j]);
}

// store one 16-byte global vector worth of channels [c, c+kVec) of row `row` into an LDS A tile
template <typename T>
__device__ __forceinline__ void lds_store_vec(T* buf, int ld, int row, int c, const Vec16& v);

template <>
__device__ __forceinline__ void lds_store_vec<half_t>(half_t* buf, int ld, int row, int c, const Vec16& v)
{
    *reinterpret_cast<Vec16*>(buf + row * ld + c) = v;
}
template <>
__device__ __forceinline__ void lds_store_vec<float>(float* buf, int ld, int row, int c, const Vec16& v)
{
    const floatx4 f = __builtin_bit_cast(floatx4, v);
#pragma unroll
    for (int j = 0; j < 4; ++j) buf[row * ld + Traits<float>::perm(c + j)] = f[j];
}

// ---------------------------------------------------------------------------------------------
// A lane's four consecutive channels [ch0, ch0+4) of pixel-row `row` <-> an LDS tile in stored order.
template <typename T>
__device__ __forceinline__ void lds_store_quad(T* buf, int ld, int row, int ch0, const floatx4& v);
template <>
__device__ __forceinline__ void lds_store_quad<half_t>(half_t* buf, int ld, int row, int ch0, const floatx4& v)
{
    typedef _Float16 half4 __attribute__((ext_vector_type(4)));
    half4 h = {Traits<half_t>::from_f(v[0]), Traits<half_t>::from_f(v[1]), Traits<half_t>::from_f(v[2]), Traits<half_t>::from_f(v[3])};
    *reinterpret_cast<half4*>(buf + row * ld + ch0) = h;
}
template <>
__device__ __forceinline__ void lds_store_quad<float>(float* buf, int ld, int row, int ch0, const floatx4& v)
{
#pragma unroll
    for (int r = 0; r < 4; ++r) buf[row * ld + Traits<float>::perm(ch0 + r)] = v[r];
}

template <typename T>
__device__ __forceinline__ floatx4 lds_load_quad(const T* buf, int ld, int row, int ch0);
template <>
__device__ __forceinline__ floatx4 lds_load_quad<half_t>(const half_t* buf, int ld, int row, int ch0)
{
    typedef _Float16 half4 __attribute__((ext_vector_type(4)));
    const half4 h = *reinterpret_cast<const half4*>(buf + row * ld + ch0);
    return floatx4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
}
template <>
__device__ __forceinline__ floatx4 lds_load_quad<float>(const float* buf, int ld, int row, int ch0)
{
    floatx4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = buf[row * ld + Traits<float>::perm(ch0 + r)];
    return v;
}

// one 16-byte vector [c, c+kVec) of row `row` of an LDS tile (stored order) <-> natural channel order
template <typename T>
__device__ __forceinline__ Vec16 lds_load_vec(const T* buf, int ld, int row, int c);
template <>
__device__ __forceinline__ Vec16 lds_load_vec<half_t>(const half_t* buf, int ld, int row, int c)
{
    return *reinterpret_cast<const Vec16*>(buf + row * ld + c);
}
template <>
__device__ __forceinline__ Vec16 lds_load_vec<float>(const float* buf, int ld, int row, int c)
{
    floatx4 f;
#pragma unroll
    for (int j = 0; j < 4; ++j) f[j] = buf[row * ld + Traits<float>::perm(c + j)];
    return __builtin_bit_cast(Vec16, f);
}

template <typename T>
__device__ __forceinline__ Vec16 pack16(const float (&f)[Traits<T>::kVec]);
template <>
__device__ __forceinline__ Vec16 pack16<half_t>(const float (&f)[8])
{
    half8 h;
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] = Traits<half_t>::from_f(f[j]);
    return __builtin_bit_cast(Vec16, h);
}
template <>
__device__ __forceinline__ Vec16 pack16<float>(const float (&f)[4])
{
    floatx4 h;
#pragma unroll
    for (int j = 0; j < 4; ++j) h[j] = f[j];
    return __builtin_bit_cast(Vec16, h);
}

// direct global store of a quad (conv epilogues): 8 bytes (f16) / 16 bytes (f32)
template <typename T>
__device__ __forceinline__ void global_store_quad(T* p, const floatx4& v);
template <>
__device__ __forceinline__ void global_store_quad<half_t>(half_t* p, const floatx4& v)
{
    typedef _Float16 half4 __attribute__((ext_vector_type(4)));
    half4 h = {Traits<half_t>::from_f(v[0]), Traits<half_t>::from_f(v[1]), Traits<half_t>::from_f(v[2]), Traits<half_t>::from_f(v[3])};
    *reinterpret_cast<half4*>(p) = h;
}
template <>
__device__ __forceinline__ void global_store_quad<float>(float* p, const floatx4& v)
{
    *reinterpret_cast<floatx4*>(p) = v;
}
