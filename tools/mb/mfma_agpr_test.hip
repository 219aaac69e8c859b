// Developer check: v_mfma_f32_32x32x16_f16 written as inline asm with the A operand in the accumulator half of the
// register file (constraint "a") against the builtin, exact small-integer data.
//   hipcc --offload-arch=gfx950 -O3 tools/mb/mfma_agpr_test.hip -o tools/mb/mfma_agpr_test
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

__global__ void k(float* out)
{
    const int lane = threadIdx.x;
    half8 a, b, a2, b2;
    for (int j = 0; j < 8; ++j) {
        a[j] = (_Float16)((lane % 32) + 1);                 // A[row][k] = row + 1
        b[j] = (_Float16)(((lane % 32) % 7) + (lane >> 5) * 8 + j);   // B[k][col]
        a2[j] = (_Float16)(((lane % 32) * 3) % 11 + j);
        b2[j] = (_Float16)((lane >> 5) + 1);
    }
    floatx16 c0;
    for (int r = 0; r < 16; ++r) c0[r] = 0.f;
    floatx16 ref = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
    ref = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b2, ref, 0, 0, 0);
    floatx16 d1, d2, d3;
    // (1) A in AGPR, B in VGPR, D in VGPR, C = 0 then accumulate
    asm volatile("s_nop 7\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(d1) : "a"(a), "v"(b));
    asm volatile("s_nop 7\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(d1) : "a"(a2), "v"(b2));
    // (2) A in AGPR, B in VGPR, D in AGPR
    asm volatile("s_nop 7\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&a"(d2) : "a"(a), "v"(b));
    asm volatile("s_nop 7\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(d2) : "a"(a2), "v"(b2));
    // (3) everything in VGPRs
    asm volatile("s_nop 7\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(d3) : "v"(a), "v"(b));
    asm volatile("s_nop 7\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(d3) : "v"(a2), "v"(b2));
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15");
    for (int r = 0; r < 16; ++r) {
        out[(0 * 16 + r) * 64 + lane] = ref[r];
        out[(1 * 16 + r) * 64 + lane] = d1[r];
        out[(2 * 16 + r) * 64 + lane] = d2[r];
        out[(3 * 16 + r) * 64 + lane] = d3[r];
    }
}

int main()
{
    float* out;
    (void)hipMalloc(&out, 4 * 16 * 64 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out);
    static float h[4 * 16 * 64];
    (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    for (int v = 1; v < 4; ++v) {
        int bad = 0, badlo = 0;
        for (int r = 0; r < 16; ++r)
            for (int l = 0; l < 64; ++l)
                if (h[(v * 16 + r) * 64 + l] != h[r * 64 + l]) {
                    ++bad;
                    if (r < 8) ++badlo;
                }
        printf("variant %d: %d of 1024 elements differ from the builtin (%d of them in registers 0-7)\n", v, bad, badlo);
    }
    printf("sample ref: %g %g %g\n", h[0], h[8 * 64 + 5], h[15 * 64 + 40]);
    return 0;
}
