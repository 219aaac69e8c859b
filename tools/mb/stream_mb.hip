// Developer micro-benchmark: the weight stream of the small-map tails.  255 workgroups x 8 waves; every workgroup reads the
// same 2 MB (252 fragments of 1 KiB per wave) through a register ring D deep, two 16x16x32 MFMAs per fragment (or none).
// rot = 0: every workgroup starts at fragment 0 (what the kernels do); rot = 1: workgroup b starts at fragment (37 b) % n
// and wraps - the same bytes per workgroup, different addresses at the same time.
//   hipcc --offload-arch=gfx950 -O3 tools/mb/stream_mb.hip -o /tmp/stream_mb && /tmp/stream_mb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int D, int NMFMA>
__global__ __launch_bounds__(512, 2) void k(const char* w, int frags, int rot, float* out, unsigned long long* cyc)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(w) + (size_t)wave * frags * 1024, 0, frags * 1024, 0x00020000);
    int pos = rot ? (int)((blockIdx.x * 37u) % (unsigned)frags) : 0;
    auto next = [&]() __attribute__((always_inline)) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, pos * 1024, 0);
        pos = pos + 1 == frags ? 0 : pos + 1;
        return __builtin_bit_cast(half8, v);
    };
    half8 ring[D], b;
    for (int j = 0; j < 8; ++j) b[j] = (_Float16)(0.001f * (lane + j));
    floatx4 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int i = 0; i < D; ++i) ring[i] = next();
    for (int f = 0; f < frags; f += D) {
#pragma unroll
        for (int i = 0; i < D; ++i) {
#pragma unroll
            for (int m = 0; m < NMFMA; ++m) acc[(i + m) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ring[i], b, acc[(i + m) & 3], 0, 0, 0);
            if (NMFMA == 0) acc[i & 3][0] += (float)ring[i][0];
            ring[i] = next();
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
    for (int i = 0; i < D; ++i) s += (float)ring[i][1];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[blockIdx.x * 2] = t1 - t0; cyc[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int D, int NMFMA>
void run(const char* w, int frags, int rot, int grid, float* out, unsigned long long* cyc)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<D, NMFMA>), dim3(grid), dim3(512), 0, 0, w, frags, rot, out, cyc);
    (void)hipEventRecord(e0, 0);
    const int n = 20;
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL((k<D, NMFMA>), dim3(grid), dim3(512), 0, 0, w, frags, rot, out, cyc);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * grid);
    (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double c = 0, r = 0;
    for (int i = 0; i < grid; ++i) { c += h[2 * i]; r += h[2 * i + 1]; }
    c /= grid; r /= grid;
    const double us = ms * 1e3 / n, bytes = 8.0 * frags * 1024;
    printf("D=%2d mfma/frag=%d rot=%d grid=%d: %6.1f us per launch; per workgroup %7.0f cycles (%.2f GHz) -> %5.1f B/clk/CU unique, %5.1f GB/s/CU\n",
           D, NMFMA, rot, grid, us, c, c / (r * 10.0) , bytes / c, bytes / (r * 10.0));
}

int main()
{
    const int frags = 252;    // 8 waves x 252 KiB = 2.06 MB: W2 + W3 + W4 of a C=384 block
    char* w; float* out; unsigned long long* cyc;
    (void)hipMalloc(&w, (size_t)8 * frags * 1024);
    (void)hipMemset(w, 0, (size_t)8 * frags * 1024);
    (void)hipMalloc(&out, 1024 * 512 * 4);
    (void)hipMalloc(&cyc, 1024 * 16);
    for (int grid : {255, 510}) {
        for (int rot = 0; rot < 2; ++rot) {
            run<8, 2>(w, frags, rot, grid, out, cyc);
            run<16, 2>(w, frags, rot, grid, out, cyc);
            run<8, 0>(w, frags, rot, grid, out, cyc);
            run<16, 0>(w, frags, rot, grid, out, cyc);
            run<8, 6>(w, frags, rot, grid, out, cyc);
        }
    }
    run<8, 2>(w, frags, 0, 64, out, cyc);
    run<8, 2>(w, frags, 0, 1, out, cyc);
    return 0;
}
