// Developer micro-benchmark: the inner GEMM loop of gemm_core.hpp in isolation (pixel tile in LDS,
// weights streamed from L2), repeated REP times per workgroup.  Reports TFLOP/s.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I opendcvc_amd/csrc tools/gemm_microbench.hip -o /tmp/gemm_mb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "gemm_core.hpp"

template <int MT, int NT, int PF, int OCC>
__global__ __launch_bounds__(256, OCC) void mb(const half8* W, float* out, int kgs, int rep)
{
    extern __shared__ __attribute__((aligned(32))) char smem[];
    half_t* X = reinterpret_cast<half_t*>(smem);
    const int ldx = kgs * 32 + 16;
    for (int i = threadIdx.x; i < MT * 16 * ldx; i += 256) X[i] = (half_t)((i % 7) * 0.125f);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int tiles[NT];
    for (int i = 0; i < NT; ++i) tiles[i] = wave + 4 * i;
    floatx4 acc[MT][NT];
    zero_acc(acc);
    for (int r = 0; r < rep; ++r) gemm_acc<half_t, MT, NT, PF>(acc, X, ldx, kgs, W, kgs, 0, tiles, lane);
    float s = 0;
    for (int m = 0; m < MT; ++m)
        for (int i = 0; i < NT; ++i) s += acc[m][i][0] + acc[m][i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MT, int NT, int PF, int OCC>
void run(const char* name, int kgs, int grid, int rep)
{
    half8* W;
    float* out;
    const size_t wb = (size_t)(4 * NT) * kgs * 64 * sizeof(half8);
    hipMalloc(&W, wb);
    hipMemset(W, 0, wb);
    hipMalloc(&out, (size_t)grid * 256 * 4);
    const size_t lds = (size_t)MT * 16 * (kgs * 32 + 16) * 2;
    hipFuncSetAttribute((const void*)mb<MT, NT, PF, OCC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int it = 0; it < 2; ++it) {
        hipEventRecord(e0);
        mb<MT, NT, PF, OCC><<<grid, 256, lds>>>(W, out, kgs, rep);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = 2.0 * grid * (MT * 16.0) * (4 * NT * 16.0) * (kgs * 32.0) * rep;
    printf("%-28s MT=%d NT=%d PF=%d occ=%d kgs=%d grid=%d: %.1f us  %.1f TFLOP/s\n", name, MT, NT, PF, OCC, kgs, grid,
           ms * 1e3, flop / ms / 1e9);
    hipFree(W);
    hipFree(out);
}

int main()
{
    run<4, 4, 2, 2>("gemm2-like 2wg/cu", 8, 512, 64);
    run<4, 4, 2, 1>("gemm2-like 1wg/cu", 8, 256, 64);
    run<4, 4, 4, 1>("gemm2-like PF4 1wg/cu", 8, 256, 64);
    run<4, 2, 4, 2>("gemm3-like 2wg/cu", 8, 512, 64);
    run<4, 4, 2, 2>("gemm4-like K=64 2wg/cu", 2, 512, 256);
    run<8, 4, 4, 1>("M=128 1wg/cu", 8, 256, 64);
    run<2, 4, 2, 2>("M=32 2wg/cu", 8, 512, 64);
    return 0;
}
