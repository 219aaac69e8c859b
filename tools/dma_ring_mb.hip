// dma_ring_mb.hip - developer micro-benchmark: how fast can every CU stream the SAME weight block (L2 resident, 896 KiB)
// into LDS, as a function of the bytes kept in flight?  Two transports:
//   mode 0: LDS-DMA (global_load_lds_dwordx4 issued from inline asm, counted vmcnt, raw barrier), ring of D slots
//   mode 1: register staging (global_load_dwordx4 -> ds_write_b128), D slots in flight in registers
// Each workgroup (256 threads) consumes slot g by reading it back from LDS (one ds_read_b128 per 1 KiB fragment and
// wave, like the MFMA consumer would) and xor-ing it into a checksum.  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int SLOT = 16384, NTHR = 256;

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int POL>
__device__ __forceinline__ void glds16p(const void* gsrc, unsigned lds_dst)
{
    unsigned keep;
    if constexpr (POL == 1)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    else if constexpr (POL == 2)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc1\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc0\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// pure fetch: no consumer, no barrier - every wave streams its quarter of each slot into its own LDS area
template <int POL>
__global__ __launch_bounds__(NTHR, 1) void dma_pure(const char* w, int nslots, unsigned* out, int unused)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned ring = (unsigned)(size_t)(smem);
    for (int g = 0; g < nslots; ++g) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const void* src = w + (size_t)g * SLOT + (wave * 4 + k) * 1024 + lane * 16;
            const unsigned dst = ring + (g % 8) * SLOT + (wave * 4 + k) * 1024;
            if constexpr (POL == 0) glds16(src, dst); else glds16p<POL>(src, dst);
        }
        asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    out[blockIdx.x * NTHR + tid] = *reinterpret_cast<unsigned*>(smem + tid * 4);
    (void)unused;
}


template <int D>
__global__ __launch_bounds__(NTHR, 1) void dma_ring(const char* w, int nslots, unsigned* out, int lds_base)
{
    const int rot = (lds_base >> 20) * blockIdx.x;     // (bits 20+ of lds_base: slot rotation per workgroup)
    lds_base &= 0xfffff;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned ring = (unsigned)(size_t)(smem) + lds_base;    // LDS byte address of the ring
    auto issue = [&](int g) {
        const int gc = ((g < nslots ? g : nslots - 1) + rot) % nslots;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            glds16(w + (size_t)gc * SLOT + (wave * 4 + k) * 1024 + lane * 16, ring + (g % D) * SLOT + (wave * 4 + k) * 1024);
    };
    for (int g = 0; g < D - 1; ++g) issue(g);
    u32x4 acc = {0, 0, 0, 0};
    for (int g = 0; g < nslots; ++g) {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(4 * (D - 2)) : "memory");
        __builtin_amdgcn_s_barrier();
        issue(g + D - 1);
        const char* s = smem + lds_base + (g % D) * SLOT + lane * 16;
#pragma unroll
        for (int f = 0; f < 16; ++f) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(s + f * 1024);
            acc ^= v;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    out[blockIdx.x * NTHR + tid] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

template <int D>
__global__ __launch_bounds__(NTHR, 1) void reg_ring(const char* w, int nslots, unsigned* out)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    u32x4 q[D][4];
    auto load = [&](u32x4 (&r)[4], int g) {
        const int gc = g < nslots ? g : nslots - 1;
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = *reinterpret_cast<const u32x4*>(w + (size_t)gc * SLOT + k * 4096 + tid * 16);
    };
#pragma unroll
    for (int d = 0; d < D; ++d) load(q[d], d + 1);
    {
        u32x4 r0[4];
        load(r0, 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) *reinterpret_cast<u32x4*>(smem + k * 4096 + tid * 16) = r0[k];
    }
    u32x4 acc = {0, 0, 0, 0};
    for (int g0 = 0; g0 < nslots; g0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int g = g0 + d;
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 4; ++k) *reinterpret_cast<u32x4*>(smem + ((g + 1) & 1) * SLOT + k * 4096 + tid * 16) = q[d][k];
            load(q[d], g + 1 + D);
            const char* s = smem + (g & 1) * SLOT + lane * 16;
#pragma unroll
            for (int f = 0; f < 16; ++f) acc ^= *reinterpret_cast<const u32x4*>(s + f * 1024);
        }
    }
    out[blockIdx.x * NTHR + tid] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

// pure VGPR streaming: no LDS, no barrier; U slots (U x 4 loads per thread) in flight, results xor-ed
template <int U>
__global__ __launch_bounds__(NTHR, 2) void reg_pure(const char* w, int nslots, unsigned* out)
{
    const int tid = threadIdx.x;
    u32x4 acc = {0, 0, 0, 0};
    for (int g0 = 0; g0 < nslots; g0 += U) {
        u32x4 r[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int k = 0; k < 4; ++k)
                r[u][k] = *reinterpret_cast<const u32x4*>(w + (size_t)((g0 + u) % nslots) * SLOT + k * 4096 + tid * 16);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc ^= r[u][k];
    }
    out[blockIdx.x * NTHR + tid] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <typename K>
float run(K kern, int lds, const char* w, int nslots, unsigned* out, int extra, bool has_extra, int iters)
{
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int w0 = 0; w0 < 2; ++w0) {
        CK(hipEventRecord(a));
        for (int i = 0; i < iters; ++i) {
            if constexpr (std::is_same<K, void (*)(const char*, int, unsigned*, int)>::value)
                hipLaunchKernelGGL(kern, dim3(255), dim3(NTHR), lds, 0, w, nslots, out, extra);
            else
                hipLaunchKernelGGL(kern, dim3(255), dim3(NTHR), lds, 0, w, nslots, out);
        }
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
    }
    (void)has_extra;
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / iters * 1e3f;
}

int main()
{
    const int nslots = 56;   // 896 KiB
    std::vector<unsigned> hw((size_t)nslots * SLOT / 4);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = (unsigned)(i * 2654435761u);
    unsigned ref = 0;
    char* dw;
    unsigned* dout;
    CK(hipMalloc(&dw, hw.size() * 4));
    CK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dout, 255 * NTHR * 4));
    // reference checksum of lane 0 of wave 0: xor over all slots / fragments of the 16 bytes at f * 1024
    for (int g = 0; g < nslots; ++g)
        for (int f = 0; f < 16; ++f)
            for (int j = 0; j < 4; ++j) ref ^= hw[((size_t)g * SLOT + f * 1024) / 4 + j];
    auto check = [&](const char* name, float us) {
        unsigned h0;
        CK(hipMemcpy(&h0, dout, 4, hipMemcpyDeviceToHost));
        printf("%-28s %7.1f us  %6.2f TB/s  %s\n", name, us, 255.0 * nslots * SLOT / us / 1e6, h0 == ref ? "ok" : "CHECKSUM MISMATCH");
    };
    check("dma D=3 (32 KB in flight)", run(dma_ring<3>, 3 * SLOT, dw, nslots, dout, 0, true, 50));
    check("dma D=4", run(dma_ring<4>, 4 * SLOT, dw, nslots, dout, 0, true, 50));
    check("dma D=6", run(dma_ring<6>, 6 * SLOT, dw, nslots, dout, 0, true, 50));
    check("dma D=8", run(dma_ring<8>, 8 * SLOT, dw, nslots, dout, 0, true, 50));
    check("dma D=6 at LDS +60 KB", run(dma_ring<6>, 6 * SLOT + 61440, dw, nslots, dout, 61440, true, 50));
    printf("(pure fetch, no consumer: checksum not meaningful)\n");
    check("pure dma default", run(dma_pure<0>, 8 * SLOT, dw, nslots, dout, 0, true, 50));
    check("pure dma nt", run(dma_pure<1>, 8 * SLOT, dw, nslots, dout, 0, true, 50));
    check("pure dma sc1", run(dma_pure<2>, 8 * SLOT, dw, nslots, dout, 0, true, 50));
    check("pure dma sc0", run(dma_pure<3>, 8 * SLOT, dw, nslots, dout, 0, true, 50));
    printf("(rotated slot order per workgroup: checksum differs by design)\n");
    check("dma D=6 rot 1", run(dma_ring<6>, 6 * SLOT, dw, nslots, dout, 1 << 20, true, 50));
    check("dma D=6 rot 7", run(dma_ring<6>, 6 * SLOT, dw, nslots, dout, 7 << 20, true, 50));
    check("dma D=6 rot 13", run(dma_ring<6>, 6 * SLOT, dw, nslots, dout, 13 << 20, true, 50));
    printf("(pure VGPR streaming, no LDS / barrier: checksum not meaningful)\n");
    check("pure vgpr U=2, 255 WGs", run(reg_pure<2>, 0, dw, nslots, dout, 0, false, 50));
    check("pure vgpr U=4, 255 WGs", run(reg_pure<4>, 0, dw, nslots, dout, 0, false, 50));
    check("pure vgpr U=8, 255 WGs", run(reg_pure<8>, 0, dw, nslots, dout, 0, false, 50));
    check("reg D=2 (+1 in LDS)", run(reg_ring<2>, 2 * SLOT, dw, nslots, dout, 0, false, 50));
    check("reg D=4", run(reg_ring<4>, 2 * SLOT, dw, nslots, dout, 0, false, 50));
    check("reg D=7", run(reg_ring<7>, 2 * SLOT, dw, nslots, dout, 0, false, 50));
    return 0;
}
