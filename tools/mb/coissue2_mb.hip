// Developer micro-benchmark (round 4): what does a SIMD do with TWO waves that each interleave MFMAs with vector work, and does an
// MFMA that is waiting for the matrix pipe hold up the OTHER wave's vector instructions?  (tools/coissue_mb.hip, round 2, found
// that a wave of back-to-back MFMAs and a second wave of gate arithmetic on the same SIMD take the SUM of their times.)
//   A: every wave runs [MFMA 32x32x16, F vector instructions, s_nop pad]; 4 waves (one per SIMD) against 8 waves (two per SIMD).
//   B: waves 0-3 run MFMAs paced by s_nop (never one waiting at the issue stage), waves 4-7 the gate arithmetic.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mb/coissue2_mb.hip -o /tmp/coissue2_mb
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int Z> __device__ __forceinline__ void pad()
{
    // Z wait states in s_nop pieces of at most 16
    if constexpr (Z >= 16) { asm volatile("s_nop 15"); pad<Z - 16>(); }
    else if constexpr (Z > 0) { asm volatile("s_nop %0" ::"n"(Z - 1)); }
}

// KIND 0: independent v_fma_f32; KIND 1: the gate's mix (v_exp_f32, v_add_f32, v_rcp_f32, v_fma_f32 in turn)
template <int F, int KIND> __device__ __forceinline__ void vec(float (&u)[16], int slot)
{
#pragma unroll
    for (int j = 0; j < F; ++j) {
        const int r = (slot * F + j) & 15;
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, 0.5" : "+v"(u[r]) : "v"(u[(r + 1) & 15]));
        else if (KIND >= 2) {
            // two-register (64-bit) vector instructions: packed fp32 arithmetic, v_mov_b64; and other candidates of the tail's epilogues
            const int r2 = r & 14;
            double& d = reinterpret_cast<double&>(u[r2]);
            const double& e = reinterpret_cast<const double&>(u[(r2 + 2) & 14]);
            if (KIND == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(d) : "v"(e));
            if (KIND == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d) : "v"(e));
            if (KIND == 4) asm volatile("v_mov_b64 %0, %1" : "+v"(d) : "v"(e));
            if (KIND == 5) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "+v"(u[r]) : "v"(u[(r + 1) & 15]), "v"(u[(r + 2) & 15]));
            if (KIND == 6) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "+v"(u[r]) : "v"(u[(r + 1) & 15]), "v"(u[(r + 2) & 15]));
            if (KIND == 7) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d) : "v"(e));
        } else {
            switch (j & 3) {
            case 0: asm volatile("v_exp_f32 %0, %0" : "+v"(u[r])); break;
            case 1: asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(u[r])); break;
            case 2: asm volatile("v_rcp_f32 %0, %0" : "+v"(u[r])); break;
            default: asm volatile("v_fma_f32 %0, %0, %1, 0.5" : "+v"(u[r]) : "v"(u[(r + 1) & 15]));
            }
        }
    }
}

template <int F, int KIND, int Z, int PRIO>
__global__ __launch_bounds__(512, 1) void ka(float* out, unsigned long long* cyc, int n)
{
    half8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x * 0.001f + j); b[j] = (_Float16)(j * 0.01f); }
    floatx16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = (float)(i + j);
    float u[16];
    for (int j = 0; j < 16; ++j) u[j] = threadIdx.x * 0.01f + j * 0.1f - 3.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (PRIO) __builtin_amdgcn_s_setprio(1);
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
            if (PRIO) __builtin_amdgcn_s_setprio(0);
            vec<F, KIND>(u, i);
            pad<Z>();
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0];
    for (int j = 0; j < 16; ++j) s += u[j];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { cyc[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2] = t0; cyc[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2 + 1] = t1; }
}

// B: role split.  mode 0: MFMA waves only, 1: vector waves only, 2: both
template <int Z, int KIND>
__global__ __launch_bounds__(512, 1) void kb(float* out, unsigned long long* cyc, int n_mfma, int n_vec, int mode)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    half8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x * 0.001f + j); b[j] = (_Float16)(j * 0.01f); }
    floatx16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = (float)(i + j);
    float u[16];
    for (int j = 0; j < 16; ++j) u[j] = threadIdx.x * 0.01f + j * 0.1f - 3.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) {
        if (mode != 1)
            for (int it = 0; it < n_mfma; ++it) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
                    pad<Z>();
                }
            }
    } else if (mode != 0) {
        for (int it = 0; it < n_vec; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) vec<8, KIND>(u, i);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0];
    for (int j = 0; j < 16; ++j) s += u[j];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { cyc[(blockIdx.x * 8 + wave) * 2] = t0; cyc[(blockIdx.x * 8 + wave) * 2 + 1] = t1; }
}

static const char* kname(int k)
{
    static const char* n[] = {"fma ", "gate", "pk_fma_f32", "pk_add_f32", "mov_b64", "fma_mix", "cvt_pk_f16", "pk_mul_f32"};
    return n[k];
}
static float* g_out;
static unsigned long long* g_cyc;

static double span(int nw)      // first start to last end over the waves of workgroup 7, in counter ticks
{
    static unsigned long long h[256 * 16];
    (void)hipMemcpy(h, g_cyc, sizeof(h), hipMemcpyDeviceToHost);
    unsigned long long lo = ~0ull, hi = 0;
    for (int w = 0; w < nw; ++w) {
        const unsigned long long s = h[(7 * 8 + w) * 2], e = h[(7 * 8 + w) * 2 + 1];
        if (e == s) continue;
        lo = s < lo ? s : lo;
        hi = e > hi ? e : hi;
    }
    return (double)(hi - lo);
}

template <int F, int KIND, int Z, int PRIO = 0>
void runa()
{
    const int n = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float us[2];
    double ticks[2];
    for (int w8 = 0; w8 < 2; ++w8) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipMemset(g_cyc, 0, 256 * 16 * 8);
            hipEventRecord(e0);
            ka<F, KIND, Z, PRIO><<<256, w8 ? 512 : 256>>>(g_out, g_cyc, n);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        us[w8] = ms * 1e3f;
        ticks[w8] = span(w8 ? 8 : 4);
    }
    // ns per MFMA of one SIMD: 4 n MFMAs per wave, one or two waves per SIMD
    printf("A  F=%2d %s pad=%2d prio=%d : 1 wave/SIMD %6.1f us = %5.2f ns per MFMA | 2 waves/SIMD %6.1f us = %5.2f ns per MFMA  (ratio %.2f; ticks %.0f / %.0f)\n",
           F, kname(KIND), Z, PRIO, us[0], us[0] * 1e3 / (4.0 * n), us[1], us[1] * 1e3 / (8.0 * n), us[1] / us[0], ticks[0], ticks[1]);
}

template <int Z, int KIND>
void runb(int n_vec)
{
    const int n = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float t[3];
    for (int mode = 0; mode < 3; ++mode) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            kb<Z, KIND><<<256, 512>>>(g_out, g_cyc, n, n_vec, mode);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        t[mode] = ms * 1e3f;
    }
    printf("B  MFMA wave pad=%2d, %s wave n=%d: MFMA alone %6.1f us, vector alone %6.1f us, both %6.1f us (sum %.1f, max %.1f)\n", Z,
           kname(KIND), n_vec, t[0], t[1], t[2], t[0] + t[1], t[0] > t[1] ? t[0] : t[1]);
}

int main(int argc, char** argv)
{
    (void)hipMalloc(&g_out, 256 * 512 * 4);
    (void)hipMalloc(&g_cyc, 256 * 16 * 8);
    if (argc > 1) {     // second set: which vector instructions share the matrix pipe?
        runb<0, 0>(1000);
        runb<0, 2>(1000);
        runb<0, 3>(1000);
        runb<0, 7>(1000);
        runb<0, 4>(1000);
        runb<0, 5>(1000);
        runb<0, 6>(1000);
        runa<4, 0, 0>();
        runa<4, 2, 0>();
        runa<4, 3, 0>();
        runa<4, 4, 0>();
        runa<4, 5, 0>();
        runa<4, 6, 0>();
        runa<8, 2, 0>();
        runa<8, 3, 0>();
        return 0;
    }
    runa<0, 0, 0>();
    runa<2, 0, 0>();
    runa<4, 0, 0>();
    runa<6, 0, 0>();
    runa<8, 0, 0>();
    runa<12, 0, 0>();
    runa<16, 0, 0>();
    runa<4, 1, 0>();
    runa<8, 1, 0>();
    runa<12, 1, 0>();
    runa<16, 1, 0>();
    runa<8, 0, 0, 1>();
    runa<8, 1, 0, 1>();
    runa<12, 1, 0, 1>();
    // a pad behind the vector work: does leaving the issue stage alone help the other wave?
    runa<8, 0, 8>();
    runa<8, 0, 16>();
    runa<8, 1, 8>();
    runa<8, 1, 16>();
    runa<0, 0, 16>();
    runa<0, 0, 24>();
    runa<0, 0, 28>();
    runa<0, 0, 32>();
    // B: paced MFMA wave beside a vector wave
    for (int nv : {1000, 2000}) {
        runb<0, 0>(nv);
        runb<16, 0>(nv);
        runb<24, 0>(nv);
        runb<28, 0>(nv);
        runb<32, 0>(nv);
        runb<0, 1>(nv);
        runb<24, 1>(nv);
        runb<28, 1>(nv);
        runb<32, 1>(nv);
    }
    return 0;
}
