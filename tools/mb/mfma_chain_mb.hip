// Developer micro-benchmark (round 4): how long does a DEPENDENT v_mfma_f32_32x32x16_f16 (same accumulator as the previous one)
// take against independent ones?  One or two waves per SIMD, NA accumulators used round-robin: NA = 1 is a single dependency
// chain (what the 32-pixel tail's W3 x o -> u phase issues: one pixel tile per wave), NA = 2, 3, 4 leave 1, 2, 3 other MFMAs
// between two that depend on each other.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mb/mfma_chain_mb.hip -o tools/mb/mfma_chain_mb
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int NA>
__global__ __launch_bounds__(512, 1) void k(float* out, int n)
{
    half8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x * 0.001f + j); b[j] = (_Float16)(j * 0.01f); }
    floatx16 acc[NA];
    for (int i = 0; i < NA; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = (float)(i + j);
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int r = 0; r < 12 / NA; ++r)
#pragma unroll
            for (int i = 0; i < NA; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NA; ++i) s += acc[i][0];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int NA>
void run(float* out, int threads)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0;
    const int n = 2000;
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(e0);
        k<NA><<<256, threads>>>(out, n);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    // 12 MFMAs per iteration and wave; waves per SIMD = threads / 256
    const double per = ms * 1e-3 / (n * 12.0 * (threads / 256));
    printf("%d wave(s) per SIMD, %d accumulator(s) round-robin: %.1f us, %.2f ns per MFMA and SIMD\n", threads / 256, NA, ms * 1e3, per * 1e9);
}

int main()
{
    float* out;
    hipMalloc(&out, 256 * 512 * sizeof(float));
    for (int threads : {256, 512}) {
        run<1>(out, threads);
        run<2>(out, threads);
        run<3>(out, threads);
        run<4>(out, threads);
    }
    return 0;
}
