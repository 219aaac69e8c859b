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
    static __device__ __forceinline__ half_t from_f(float v) { return (half_t)v; }
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
};

// acc[m][i] += A[m-tile m][K] * W[n-tile tiles[i]][K]^T over `kgs` reduction groups.
//   A      : LDS, row-major [16*MT][lda] in stored (perm) order, groups 0..kgs-1
//   Wp     : packed weights, fragment (tile, group g) at Wp[(tile * kgs_total + g) * 64 + lane]
//   kg0    : first group of W to use (A group g pairs with W group kg0 + g)
template <typename T, int MT, int NT>
__device__ __forceinline__ void gemm_acc(floatx4 (&acc)[MT][NT], const T* A, int lda, int kgs,
                                         const typename Traits<T>::frag_t* __restrict__ Wp,
                                         int kgs_total, int kg0, const int (&tiles)[NT], int lane)
{
    using frag_t = typename Traits<T>::frag_t;
    const int r = lane & 15, q = lane >> 4;
    const T* a_base = A + r * lda + q * 8;
    const frag_t* w_base[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) w_base[i] = Wp + ((size_t)tiles[i] * kgs_total + kg0) * 64 + lane;

    frag_t bcur[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) bcur[i] = w_base[i][0];

    for (int g = 0; g < kgs; ++g) {
        frag_t bnext[NT];
        const int gn = (g + 1 < kgs) ? g + 1 : g;
#pragma unroll
        for (int i = 0; i < NT; ++i) bnext[i] = w_base[i][(size_t)gn * 64];
        frag_t a[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) a[m] = *reinterpret_cast<const frag_t*>(a_base + m * 16 * lda + g * KG);
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m][i] = Traits<T>::mma(a[m], bcur[i], acc[m][i]);
#pragma unroll
        for (int i = 0; i < NT; ++i) bcur[i] = bnext[i];
    }
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
struct alignas(16) Vec16 {
    uint32_t w[4];
};

template <typename T>
__device__ __forceinline__ void unpack16(const Vec16& v, float (&f)[Traits<T>::kVec]);

template <>
__device__ __forceinline__ void unpack16<half_t>(const Vec16& v, float (&f)[8])
{
    const half_t* h = reinterpret_cast<const half_t*>(&v);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (float)h[j];
}
template <>
__device__ __forceinline__ void unpack16<float>(const Vec16& v, float (&f)[4])
{
    const float* h = reinterpret_cast<const float*>(&v);
#pragma unroll
    for (int j = 0; j < 4; ++j) f[j] = h[j];
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
    const float* f = reinterpret_cast<const float*>(&v);
#pragma unroll
    for (int j = 0; j < 4; ++j) buf[row * ld + Traits<float>::perm(c + j)] = f[j];
}
