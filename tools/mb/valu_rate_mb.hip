// Developer micro-benchmark: issue cost (cycles per wave-instruction, one wave per SIMD) of the vector instructions the
// depthwise stage can be built from.   hipcc --offload-arch=gfx950 -O3 tools/mb/valu_rate_mb.hip -o /tmp/valu_rate_mb
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(x) x x x x x x x x x x x x x x x x
template <int MODE>
__global__ __launch_bounds__(1024, 1) void k(float* out, unsigned long long* cyc, int n)
{
    typedef _Float16 half8 __attribute__((ext_vector_type(8)));
    typedef float floatx16 __attribute__((ext_vector_type(16)));
    typedef float floatx4 __attribute__((ext_vector_type(4)));
    half8 ha, hb;
    for (int j = 0; j < 8; ++j) { ha[j] = (_Float16)(threadIdx.x * 0.001f + j); hb[j] = (_Float16)(j * 0.01f + 0.3f); }
    floatx16 c32[4];
    floatx4 c16[4];
    for (int i = 0; i < 4; ++i) { for (int j = 0; j < 16; ++j) c32[i][j] = j; for (int j = 0; j < 4; ++j) c16[i][j] = j; }
    float acc[16];
    unsigned a = threadIdx.x * 0x3c003c01u, b = 0x3c003800u + threadIdx.x;
    for (int i = 0; i < 16; ++i) acc[i] = i;
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "+v"(acc[i]) : "v"(a), "v"(b));
            if (MODE == 1) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
            if (MODE == 2) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
            if (MODE == 3) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(acc[i]) : "v"(a));
            if (MODE == 4) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
            if (MODE == 5) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(acc[i]) : "v"(a), "v"(b));
            if (MODE == 6) asm volatile("v_pk_add_f16 %0, %1, %2" : "=v"(acc[i]) : "v"(a), "v"(b));
            if (MODE == 8) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c32[i & 3]) : "v"(ha), "v"(hb));
            if (MODE == 9) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c16[i & 3]) : "v"(ha), "v"(hb));
            if (MODE == 10) asm volatile("v_exp_f32 %0, %1" : "=v"(acc[i]) : "v"(a));
            if (MODE == 7) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 4; ++i) s += c32[i][0] + c16[i][0];
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * 1024 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { const int w = threadIdx.x >> 6; cyc[(blockIdx.x * 16 + w) * 2] = t0; cyc[(blockIdx.x * 16 + w) * 2 + 1] = t1; }
    if (threadIdx.x == 0) cyc[256 * 32 + blockIdx.x] = (t1 - t0) * 100 / (r1 - r0);
}

template <int MODE>
void run(const char* name, float* out, unsigned long long* cyc, int threads = 256)
{
    const int n = 2000;
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, n);
    (void)hipDeviceSynchronize();
    static unsigned long long h[256 * 32 + 256];
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const int nw = threads / 64, b = 7;
    unsigned long long lo = ~0ull, hi = 0;
    for (int w = 0; w < nw; ++w) {
        lo = h[(b * 16 + w) * 2] < lo ? h[(b * 16 + w) * 2] : lo;
        hi = h[(b * 16 + w) * 2 + 1] > hi ? h[(b * 16 + w) * 2 + 1] : hi;
    }
    // all waves of the workgroup, first start to last end, per instruction of ONE SIMD's waves (nw / 4 waves per SIMD)
    printf("%-18s %.2f cycles per wave-instruction on one SIMD (%d waves per SIMD; wave 0 alone: %.2f); clock %.0f MHz\n", name,
           (double)(hi - lo) / (16.0 * n * (nw / 4)), nw / 4, (double)(h[(b * 16) * 2 + 1] - h[(b * 16) * 2]) / (16.0 * n),
           (double)h[256 * 32 + b]);
}

int main()
{
    float* out;
    unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 1024 * 4);
    (void)hipMalloc(&cyc, (256 * 32 + 256) * 8);
    run<0>("v_fma_mix_f32", out, cyc);
    run<1>("v_fma_f32", out, cyc);
    run<2>("v_pk_fma_f16", out, cyc);
    run<3>("v_cvt_f32_f16", out, cyc);
    run<4>("v_dot2_f32_f16", out, cyc);
    run<5>("v_cvt_pk_f16_f32", out, cyc);
    run<6>("v_pk_add_f16", out, cyc);
    run<7>("v_dot2c_f32_f16", out, cyc);
    run<8>("mfma_32x32x16_f16", out, cyc);
    run<9>("mfma_16x16x32_f16", out, cyc);
    run<10>("v_exp_f32", out, cyc);
    for (int th : {512, 1024}) {
        printf("-- %d waves per SIMD (time of wave 0; all waves run the same loop)\n", th / 256);
        run<1>("v_fma_f32", out, cyc, th);
        run<0>("v_fma_mix_f32", out, cyc, th);
        run<10>("v_exp_f32", out, cyc, th);
        run<8>("mfma_32x32x16_f16", out, cyc, th);
    }
    return 0;
}
