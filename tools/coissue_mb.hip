// Developer micro-benchmark: do MFMA work of one wave and transcendental-heavy VALU work of ANOTHER wave on the same
// SIMD overlap?  One 8-wave workgroup per CU (waves w and w+4 share a SIMD); waves 0-3 run an MFMA chain on
// registers, waves 4-7 evaluate the fp16-mode gate u / (1 + 2^u).  Modes: MFMA only, VALU only, both.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/coissue_mb.hip -o tools/coissue_mb
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int SHAPE, int PM = 0, int PV = 0>   // 0: 16x16x32, 1: 32x32x16; wave priorities of the MFMA / VALU waves
__global__ __launch_bounds__(512, 1) void k(float* out, int n_mfma, int n_valu, int mode)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float s = 0;
    if (wave < 4) {
        if (mode == 1) return;
        __builtin_amdgcn_s_setprio(PM);
        half8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x * 0.001f + j); b[j] = (_Float16)(j * 0.01f); }
        if (SHAPE == 0) {
            floatx4 acc[8];
            for (int i = 0; i < 8; ++i) acc[i] = floatx4{0, 0, 0, 0};
            for (int it = 0; it < n_mfma; ++it)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
            for (int i = 0; i < 8; ++i) s += acc[i][0];
        } else {
            floatx16 acc[4];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 16; ++j) acc[i][j] = 0;
            for (int it = 0; it < n_mfma; ++it)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
            for (int i = 0; i < 4; ++i) s += acc[i][0];
        }
    } else {
        if (mode == 0) return;
        __builtin_amdgcn_s_setprio(PV);
        float u[16];
        for (int j = 0; j < 16; ++j) u[j] = threadIdx.x * 0.01f + j * 0.1f - 3.f;
        for (int it = 0; it < n_valu; ++it)
#pragma unroll
            for (int j = 0; j < 16; ++j) u[j] = u[j] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u[j])) + 0.25f;
        for (int j = 0; j < 16; ++j) s += u[j];
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

// one wave per SIMD, MFMA and gates interleaved in ONE instruction stream: G gates per 8 MFMAs.
// (MFMAs through inline asm so that hipcc neither merges the identical products nor rotates the accumulators.)
template <int G>
__global__ __launch_bounds__(256, 1) void k1(float* out, int n)
{
    half8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x * 0.001f + j); b[j] = (_Float16)(j * 0.01f); }
    floatx4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = floatx4{0, 0, 0, (float)i};
    float u[G > 0 ? G : 1];
    for (int j = 0; j < G; ++j) u[j] = threadIdx.x * 0.01f + j * 0.1f - 3.f;
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
            for (int j = 0; j < G; ++j)
                if (j % 8 == i || G >= 8) {
                    if (G < 8 || (j / (G / 8)) == i) {
                        float e, r;
                        asm volatile("v_exp_f32 %0, %1" : "=v"(e) : "v"(u[j]));
                        asm volatile("v_add_f32 %0, 1.0, %1" : "=v"(e) : "v"(e));
                        asm volatile("v_rcp_f32 %0, %1" : "=v"(r) : "v"(e));
                        asm volatile("v_fma_f32 %0, %1, %2, 0.5" : "=v"(u[j]) : "v"(u[j]), "v"(r));
                    }
                }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0];
    for (int j = 0; j < G; ++j) s += u[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int G>
void run1(float* out)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0;
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(e0);
        k1<G><<<256, 256>>>(out, 4000);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    printf("same wave, %2d gates per 8 MFMAs: %.1f us\n", G, ms * 1e3f);
}

// same wave: every MFMA followed by F independent plain VALU instructions (v_add_f32) or transcendental ones
template <int SHAPE, int F, int TRANS>
__global__ __launch_bounds__(256, 1) void k2(float* out, int n)
{
    half8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x * 0.001f + j); b[j] = (_Float16)(j * 0.01f); }
    floatx4 acc4[8];
    floatx16 acc16[4];
    for (int i = 0; i < 8; ++i) acc4[i] = floatx4{0, 0, 0, (float)i};
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) acc16[i][j] = (float)(i + j);
    float u[F > 0 ? F : 1];
    for (int j = 0; j < F; ++j) u[j] = threadIdx.x * 0.01f + j * 0.1f;
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < (SHAPE == 0 ? 8 : 4); ++i) {
            if (SHAPE == 0)
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc4[i]) : "v"(a), "v"(b));
            else
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc16[i]) : "v"(a), "v"(b));
#pragma unroll
            for (int j = 0; j < F; ++j) {
                if (TRANS)
                    asm volatile("v_exp_f32 %0, %0" : "+v"(u[j]));
                else
                    asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(u[j]));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc4[i][0];
    for (int i = 0; i < 4; ++i) s += acc16[i][0];
    for (int j = 0; j < F; ++j) s += u[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int SHAPE, int F, int TRANS>
void run2(float* out)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0;
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(e0);
        k2<SHAPE, F, TRANS><<<256, 256>>>(out, 4000);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    printf("same wave %s, %d %s after every MFMA: %.1f us\n", SHAPE == 0 ? "16x16x32" : "32x32x16", F, TRANS ? "v_exp_f32" : "v_add_f32", ms * 1e3f);
}

template <int SHAPE, int PM = 0, int PV = 0>
float run(float* out, int nm, int nv, int mode)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0;
    for (int it = 0; it < 2; ++it) {
        hipEventRecord(e0);
        k<SHAPE, PM, PV><<<256, 512>>>(out, nm, nv, mode);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    return ms * 1e3f;
}

int main()
{
    float* out;
    hipMalloc(&out, 256 * 512 * 4);
    const int nm = 4000;          // x8 MFMAs of 16 cycles (or x4 of 32) = 512 K MFMA-pipe cycles per wave
    for (int nv : {1000, 2000, 4000}) {   // x16 gates x 4 instr
        for (int shape = 0; shape < 2; ++shape) {
            float t[3];
            for (int mode = 0; mode < 3; ++mode) t[mode] = shape == 0 ? run<0>(out, nm, nv, mode) : run<1>(out, nm, nv, mode);
            printf("%s  n_valu=%d: MFMA alone %.1f us, VALU alone %.1f us, both %.1f us (sum %.1f, max %.1f)\n",
                   shape == 0 ? "16x16x32" : "32x32x16", nv, t[0], t[1], t[2], t[0] + t[1], t[0] > t[1] ? t[0] : t[1]);
        }
    }
    // the same with wave priorities (s_setprio): MFMA waves raised, then VALU waves raised
    for (int nv : {1000, 2000}) {
        printf("16x16x32 n_valu=%d both: prio(mfma,valu) (0,0) %.1f  (1,0) %.1f  (3,0) %.1f  (0,1) %.1f us\n", nv, run<0, 0, 0>(out, nm, nv, 2),
               run<0, 1, 0>(out, nm, nv, 2), run<0, 3, 0>(out, nm, nv, 2), run<0, 0, 1>(out, nm, nv, 2));
        printf("32x32x16 n_valu=%d both: prio(mfma,valu) (0,0) %.1f  (1,0) %.1f  (3,0) %.1f  (0,1) %.1f us\n", nv, run<1, 0, 0>(out, nm, nv, 2),
               run<1, 1, 0>(out, nm, nv, 2), run<1, 3, 0>(out, nm, nv, 2), run<1, 0, 1>(out, nm, nv, 2));
    }
    run2<0, 0, 0>(out);
    run2<0, 1, 0>(out);
    run2<0, 2, 0>(out);
    run2<0, 4, 0>(out);
    run2<0, 1, 1>(out);
    run2<0, 2, 1>(out);
    run2<1, 0, 0>(out);
    run2<1, 2, 0>(out);
    run2<1, 4, 0>(out);
    run2<1, 6, 0>(out);
    run2<1, 8, 0>(out);
    run2<1, 2, 1>(out);
    run2<1, 4, 1>(out);
    run1<0>(out);
    run1<1>(out);
    run1<2>(out);
    run1<4>(out);
    run1<8>(out);
    run1<16>(out);
    return 0;
}
