// dcb_ps.hpp - "pixel-stationary" fused DepthConvBlock tail for gfx950, fp16 storage / fp32 accumulate.
//
//   d = dw3x3(a) + bd;  o = (W2 d + b2) + x';  v = g(u_lo) + g(u_hi), u = W3 o + b3;
//   r = ((W4 v + b4) + o);  out = (r [+ x']) [* q];   [a' = g(W1' r + b1')  (next block's fused head)]
// (reference: DepthConvBlock.forward_torch, src/layers/layers.py:92-106; impl.cpp:53-121 runs it as 8 launches)
//
// Decomposition (differs from the channel-split kernels of dcvc_nn.hip, which stay for fp32 and small maps):
//   * a 256-thread workgroup (4 waves, ONE per SIMD, up to 512 VGPRs each) owns a 8 x 16 = 128-pixel tile; each
//     wave owns 32 of those pixels (two tile rows) and ALL channels.  v_mfma_f32_32x32x16_f16 with the weights as
//     the A operand (32 output channels x 16 k) and the wave's 32 pixels as the B operand (16 k x 32 pixels).
//   * the whole activation chain of a pixel stays in ITS wave's registers: the 32 x 32 accumulator tile of one
//     GEMM (lane = pixel, 16 registers = channels) is, after the epilogue and a cvt to f16, directly the B
//     fragment pair of the next GEMM (cdna_hip_programming.md, "An accumulator tile as the next MFMA's operand").
//     The row permutation that idiom needs (swap of channel bits 2 and 3 inside a 32-tile) is applied to the ROWS
//     of every packed weight tile on the host, so all reductions run over channels in natural order.  No LDS
//     round trip, no barrier and no cross-lane traffic between depthwise -> W2 -> FFN -> output.
//   * the weights (7 C^2 halves per block, the same for every workgroup, L2 resident) are streamed through a
//     three-buffer LDS ring in the exact order the waves consume them, one "slot" of C/16 fragments (C/16 KiB) per
//     barrier: at step g every thread requests its 16-byte pieces of slot g+2+DS from L2 (DS slots stay in flight in
//     registers: an L2 round trip is ~1 us under load, several slots long), writes slot g+2 to LDS and the four waves
//     run slot g's MFMAs while already reading slot g+1's first fragments (the third buffer is what lets the
//     fragment reads run one group ahead across the barrier).  A fragment is read from L2 once per 128 pixels (the
//     channel-split kernels: once per 64) and feeds one MFMA (32 cycles) per wave from LDS (1 KiB per MFMA per SIMD
//     = 128 B/clk/CU, half the LDS read rate).
//   * depthwise 3x3: the activation tile + halo goes through LDS in 64-channel slabs (two buffers); a lane computes
//     the 8 channels x 9 taps of its own pixel for one k-step straight into a B fragment, one k-step ahead of the
//     W2 MFMAs that consume it (the depthwise VALU work runs underneath those MFMAs).
//   * FFN: the C -> 4C product is taken 32 v-channels (one u_lo + one u_hi tile) at a time; the gate of tile j runs
//     on the VALU while the matrix pipe does tile j+1's 32 MFMAs, then tile j's 2C -> C slice (16 MFMAs for
//     C = 256) accumulates into the C/32 output tiles that stay in registers for the whole block.
// Registers (C = 256): o fragments 64 + output accumulators 128 + two u tile pairs 64 + weight fragments ~32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>
#include <utility>

#include "gemm_core.hpp"

namespace ps {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

constexpr int TH = 8, TW = 16, HALO_W = TW + 2, HALO_H = TH + 2, HALO = HALO_W * HALO_H;   // 180 halo pixels
constexpr int SLAB = 64;                       // channels per depthwise slab
constexpr int SLAB_BYTES = HALO * SLAB * 2;    // 23 040
constexpr int NTHR = 256;

// compile-time loop: f(std::integral_constant<int, I>{}) for I = 0 .. N-1
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>)
{
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f)
{
    static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}
template <int V>
using ic = std::integral_constant<int, V>;

// sigma: row m of a packed 32-row weight tile holds channel sigma(m) of the tile (bits 2 and 3 swapped), so that the
// D registers 8s..8s+7 of lane (pixel, h) are the 8 consecutive channels 16 s + 8 h + {0..7}
__host__ __device__ constexpr int sigma(int m) { return (m & ~12) | ((m & 4) << 1) | ((m & 8) >> 1); }

template <int C>
struct Cfg {
    static_assert(C % 64 == 0, "width must be padded to a multiple of 64");
    static constexpr int NT = C / 32;              // 32-channel tiles of a C-wide tensor
    static constexpr int KS = C / 16;              // k-steps of 16 over C channels  ( = v tiles of 32 over 2C )
    static constexpr int SLOT_FRAGS = C / 16;      // 1 KiB fragments per ring slot
    static constexpr int SLOT_BYTES = SLOT_FRAGS * 1024;
    static constexpr int R = SLOT_BYTES / (NTHR * 16);   // 16-byte pieces per thread per slot
    static constexpr int NSLAB = C / SLAB;
    static constexpr int TAIL_SLOTS = NT + 3 * KS;  // W2 (k-outer, 2 k-steps per slot) + per v tile: W3 lo, W3 hi, W4
    static constexpr int HEAD_SLOTS = NT;           // W1 tile-outer (one output tile per slot)
    // LDS: ring | halo slabs | tables
    static constexpr int RING_OFF = 0;
    static constexpr int NBUF = 6;                  // LDS ring buffers (slot g lives in buffer g % 6), filled by LDS-DMA
    static constexpr int HALO_OFF = NBUF * SLOT_BYTES;
    static constexpr int TBL_OFF = HALO_OFF + 2 * SLAB_BYTES;
    // tables: wd [9][C] f16 | bd [C] f32 | b2 [C] | b3 [4C] | b4 [C] | nb1 [C]
    static constexpr int T_WD = 0, T_BD = 18 * C, T_B2 = T_BD + 4 * C, T_B3 = T_B2 + 4 * C, T_B4 = T_B3 + 16 * C,
                         T_NB1 = T_B4 + 4 * C, TBL_BYTES = T_NB1 + 4 * C;
    static constexpr int STAGE_OFF = HALO_OFF;      // output staging reuses the halo slabs (+ tables stay)
    static constexpr int LDS_BYTES = TBL_OFF + TBL_BYTES;
    // the output tile is staged through LDS 64 channels at a time (128 px x 128 B = 16 KiB, two buffers)
    static_assert(2 * 16384 <= 2 * SLAB_BYTES, "staging buffers must fit in the halo region");
};

struct Params {
    const void* a;        // activation of the first conv (dcb_head_kernel / previous block's fused head), HWC
    long lda;
    const void* ident;    // x' (block input or adaptor output)
    long ldi;
    int H, W;
    int c_log;
    const void* stream;   // packed tail weights in consumption order (Cfg::TAIL_SLOTS slots)
    const void* tables;   // packed wd | bd | b2 | b3 | b4 (Cfg::T_NB1 bytes)
    int shortcut;
    const float* q;
    void* out;
    long ldo;
    const void* nstream;  // next block's packed first conv (Cfg::HEAD_SLOTS slots) or NULL
    const float* nb1;
    void* na_out;
    long nlda;
    unsigned long long* stamps;   // diagnostic build (-DDCVC_DIAG) only: 8 cycle stamps per workgroup
};

#ifdef DCVC_DIAG
#define PS_STAMP(k)                                                                \
    do {                                                                           \
        if (p.stamps && threadIdx.x == 0) {                                        \
            __builtin_amdgcn_sched_barrier(0);                                     \
            p.stamps[(size_t)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime(); \
            __builtin_amdgcn_sched_barrier(0);                                     \
        }                                                                          \
    } while (0)
#else
#define PS_STAMP(k) \
    do {            \
    } while (0)
#endif

__device__ __forceinline__ floatx16 mfma32(const half8& a, const half8& b, const floatx16& c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float gate1(float up) { return Traits<half_t>::gate(up); }

// one LDS-DMA piece: 64 lanes x 16 bytes from per-lane global addresses to LDS bytes [lds_dst, lds_dst + 1024).
// M0 carries the LDS address and is compiler-reserved: saved and restored inside the statement
// (cdna_hip_programming.md section 5.7).
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

// 8 consecutive fp32 values -> one B fragment (8 halves)
__device__ __forceinline__ half8 pack8(const float (&v)[8])
{
    half8 h;
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] = (half_t)v[j];
    return h;
}

template <int C>
__global__ __launch_bounds__(NTHR, 1) void dcb_tail_ps_kernel(Params p)
{
    using CF = Cfg<C>;
    constexpr int NT = CF::NT, KS = CF::KS, R = CF::R, SB = CF::SLOT_BYTES, NBUF = CF::NBUF;
    constexpr int GS = NT / 2;                                 // fragments per read group: a slot = 4 groups
    static_assert(NT % 2 == 0, "fragment groups are half a k-step / a quarter of a u tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pp = lane & 31, h = lane >> 5;                  // the lane's pixel inside the wave, k / row half
    const int tiles_x = (p.W + TW - 1) / TW;
    const int ty0 = (blockIdx.x / tiles_x) * TH, tx0 = (blockIdx.x % tiles_x) * TW;
    const int ly = 2 * wave + (pp >> 4), lx = pp & 15;        // pixel inside the 8 x 16 tile
    const int gy = ty0 + ly, gx = tx0 + lx;
    const bool pvalid = gy < p.H && gx < p.W;
    const long gpix = (long)(pvalid ? gy : 0) * p.W + (pvalid ? gx : 0);
    const half_t* a = reinterpret_cast<const half_t*>(p.a);
    const half_t* ident = reinterpret_cast<const half_t*>(p.ident);
    char* ring = smem + CF::RING_OFF;
    char* halo = smem + CF::HALO_OFF;
    const char* tbl = smem + CF::TBL_OFF;

    // ---- weight ring -------------------------------------------------------------------------------------
    // LDS-DMA (global_load_lds_dwordx4: L2 -> LDS without passing through registers; measured on this chip, every CU
    // streaming the same 896 KiB: 27 TB/s for bare DMA against ~10 TB/s through VGPRs, tools/dma_ring_mb.hip).
    // Slot g is requested at step g - (NBUF - 1) into buffer g % NBUF, waited for (counted vmcnt: its requests are
    // older than the NBUF - 3 slots issued after it) before the barrier of step g - 1 and consumed at step g, the
    // fragment reads running one group ahead of the MFMAs across that barrier.  The DMA is issued from inline asm:
    // hipcc then does not know that LDS is being written behind its back and leaves the ds_reads their counted
    // lgkmcnt waits (with the builtin it drains vmcnt before every LDS read that may alias a transfer in flight).
    const int total_slots = CF::TAIL_SLOTS + (p.nstream ? CF::HEAD_SLOTS : 0);
    [[maybe_unused]] unsigned long long t_lds = 0, t_bar = 0, t_commit = 0;   // diagnostic build: where ring_step waits
    const unsigned lds_base = (unsigned)(size_t)smem;         // LDS byte address of the dynamic segment
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto ring_issue = [&](auto buf, int g) {                  // slot g -> buffer B: R one-KiB pieces per wave
        constexpr int B = decltype(buf)::value;
        const int gc = g < total_slots ? g : total_slots - 1;      // (clamped: the requests past the end are harmless)
        const char* base = gc < CF::TAIL_SLOTS ? reinterpret_cast<const char*>(p.stream) + (size_t)gc * SB
                                               : reinterpret_cast<const char*>(p.nstream) + (size_t)(gc - CF::TAIL_SLOTS) * SB;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int piece = wave_u * R + k;
            glds16(base + piece * 1024 + lane * 16, lds_base + CF::RING_OFF + B * SB + piece * 1024);
        }
    };
    // one piece of a slot (spread over the four MFMA groups of a step: a DMA takes ~50 issue cycles, four in a row
    // drain the matrix pipe)
    const char* issue_base = nullptr;
    auto ring_issue_begin = [&](int g) {
        const int gc = g < total_slots ? g : total_slots - 1;
        issue_base = gc < CF::TAIL_SLOTS ? reinterpret_cast<const char*>(p.stream) + (size_t)gc * SB
                                         : reinterpret_cast<const char*>(p.nstream) + (size_t)(gc - CF::TAIL_SLOTS) * SB;
    };
    auto ring_issue_piece = [&](auto buf, auto kc) {
        constexpr int B = decltype(buf)::value, k0 = decltype(kc)::value;
#pragma unroll
        for (int k = k0; k < R; k += 4) {
            const int piece = wave_u * R + k;
            glds16(issue_base + piece * 1024 + lane * 16, lds_base + CF::RING_OFF + B * SB + piece * 1024);
        }
    };
    // step g (G = g as far as the phases are concerned): slot g + 1 landed, barrier, slot g + NBUF - 1 requested
    auto ring_step = [&](auto gph, int g) {
#ifdef DCVC_DIAG
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long ta = __builtin_amdgcn_s_memtime();
#endif
        // own requests of slot g + 1 have landed; own LDS reads of everything but the group requested a moment ago
        // (the two groups for the other side of the barrier) have returned, in particular those of slot g - 1, whose buffer is
        // overwritten after the barrier
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(%1)" ::"n"((NBUF - 3) * R), "n"(2 * GS) : "memory");
#ifdef DCVC_DIAG
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long tb = __builtin_amdgcn_s_memtime();
#endif
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#ifdef DCVC_DIAG
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long tc = __builtin_amdgcn_s_memtime();
#endif
        ring_issue_begin(g + NBUF - 1);           // (its pieces are issued by slot_groups, one per MFMA group)
#ifdef DCVC_DIAG
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long td = __builtin_amdgcn_s_memtime();
        t_lds += tb - ta;
        t_bar += tc - tb;
        t_commit += td - tc;
#endif
    };
    // fragment groups: GS fragments (a quarter of a slot) at a time in a three-deep register ring, read TWO groups
    // (2 GS MFMAs = 256 cycles at C = 256) ahead of the MFMAs that use them, across the slot boundary too: with four
    // waves reading 1 KiB per MFMA and the DMA writing next to them an LDS read takes a few hundred cycles to return.
    // Global group index Q = 4 * slot + (0..3); buffer Q % 3.
    half8 F0[GS], F1[GS], F2[GS];
    auto read_q = [&](auto qc) {
        constexpr int Q = decltype(qc)::value, B = (Q / 4) % NBUF, grp = Q % 4;
        const char* src = ring + B * SB + grp * GS * 1024 + lane * 16;
#pragma unroll
        for (int t = 0; t < GS; ++t) {
            const half8 v = *reinterpret_cast<const half8*>(src + t * 1024);
            if constexpr (Q % 3 == 0) F0[t] = v; else if constexpr (Q % 3 == 1) F1[t] = v; else F2[t] = v;
        }
    };
    // the four groups of slot G: request group Q + 2, then body(j, fragments of group Q)
    auto slot_groups = [&](auto gph, auto&& body) {
        constexpr int G = decltype(gph)::value;
        static_for<4>([&](auto jc) {
            constexpr int Q = 4 * G + decltype(jc)::value;
            ring_issue_piece(ic<(G + NBUF - 1) % NBUF>{}, jc);
            read_q(ic<Q + 2>{});
            if constexpr (Q % 3 == 0) body(jc, F0); else if constexpr (Q % 3 == 1) body(jc, F1); else body(jc, F2);
        });
    };

    // ---- halo slabs ------------------------------------------------------------------------------------------
    // slab s = channels [64 s, 64 s + 64) of the 10 x 18 halo pixels; pixel n at byte n * 128, its 16-byte chunk c
    // (8 channels) in slot c ^ ((n >> 1) & 7) (keeps the 16-lane groups of ds_read_b128 on distinct banks)
    constexpr int HPIECES = (HALO * 8 + NTHR - 1) / NTHR;     // 16-byte pieces per thread per slab (6)
    Vec16 hreg[HPIECES];
    auto halo_load = [&](int s) {
        const int sc = s < CF::NSLAB ? s : CF::NSLAB - 1;
#pragma unroll
        for (int k = 0; k < HPIECES; ++k) {
            const int it = tid + k * NTHR;
            const int n = it >> 3, slot = it & 7, c = slot ^ ((n >> 1) & 7);
            const int y = ty0 - 1 + n / HALO_W, x = tx0 - 1 + n % HALO_W;
            Vec16 v = VEC16_ZERO;
            if (it < HALO * 8 && y >= 0 && y < p.H && x >= 0 && x < p.W)
                v = *reinterpret_cast<const Vec16*>(a + ((long)y * p.W + x) * p.lda + sc * SLAB + c * 8);
            hreg[k] = v;
        }
    };
    auto halo_commit = [&](int s) {
        char* dst = halo + (s & 1) * SLAB_BYTES;
#pragma unroll
        for (int k = 0; k < HPIECES; ++k) {
            const int it = tid + k * NTHR;
            if (it < HALO * 8) *reinterpret_cast<Vec16*>(dst + it * 16) = hreg[k];
        }
    };

    PS_STAMP(0);
    // ---- prologue ------------------------------------------------------------------------------------------
    // the first NBUF - 1 ring slots are requested right away; halo slab 0 and the tables follow through registers
    static_for<NBUF - 1>([&](auto d) { ring_issue(d, decltype(d)::value); });
    halo_load(0);
    {   // tables -> LDS (plain copy, 16-byte pieces)
        const Vec16* src = reinterpret_cast<const Vec16*>(p.tables);
        Vec16* dst = reinterpret_cast<Vec16*>(smem + CF::TBL_OFF);
        for (int it = tid; it < CF::T_NB1 / 16; it += NTHR) dst[it] = src[it];
        if (p.nstream) {
            const Vec16* nb = reinterpret_cast<const Vec16*>(p.nb1);
            Vec16* nd = reinterpret_cast<Vec16*>(smem + CF::TBL_OFF + CF::T_NB1);
            for (int it = tid; it < C * 4 / 16; it += NTHR) nd[it] = nb[it];
        }
    }
    halo_commit(0);
    halo_load(1);
    Vec16 idf[KS];      // identity fragments of this lane's pixel: idf[ks] = x'[pixel][16 ks + 8 h .. +7] (requested below)

    // depthwise of k-step ks for this lane's pixel: 8 channels x 9 taps -> B fragment, in two halves: the 20 LDS reads
    // are issued one MFMA group before the 72 multiply-adds that use them (left to itself hipcc keeps each tap's
    // read -> wait -> 8 fma together and pays the LDS latency nine times per k-step)
    const int hn0 = ly * HALO_W + lx;                          // halo index of tap (0, 0)
    Vec16 tap_a[9];
    int tap_ks = 0;
    auto dw_load = [&](int ks) {
        tap_ks = ks;
        const char* hb = halo + ((ks >> 2) & 1) * SLAB_BYTES;
        const int c = 2 * (ks & 3) + h;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int n = hn0 + ky * HALO_W + kx;
                tap_a[ky * 3 + kx] = *reinterpret_cast<const Vec16*>(hb + n * 128 + ((c ^ ((n >> 1) & 7)) << 4));
            }
    };
    auto dw_math = [&]() -> half8 {      // (the tap weights and the bias - broadcast reads of the tables - are read here)
        const int ks = tap_ks;
        Vec16 tap_w[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) tap_w[t] = *reinterpret_cast<const Vec16*>(tbl + CF::T_WD + (t * C + ks * 16 + h * 8) * 2);
        const floatx4 tap_b0 = *reinterpret_cast<const floatx4*>(tbl + CF::T_BD + (ks * 16 + h * 8) * 4);
        const floatx4 tap_b1 = *reinterpret_cast<const floatx4*>(tbl + CF::T_BD + (ks * 16 + h * 8 + 4) * 4);
        float s[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) fma_vec16<half_t>(tap_a[t], tap_w[t], s);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s[j] = s[j] + tap_b0[j];
            s[j + 4] = s[j + 4] + tap_b1[j];
        }
        return pack8(s);
    };
    auto bias16 = [&](int off_bytes, int ch0) -> floatx16 {   // table floats ch0 + 16 s + 8 h + j  ->  D register 8 s + j
        floatx16 r;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int q4 = 0; q4 < 2; ++q4) {
                const floatx4 v = *reinterpret_cast<const floatx4*>(tbl + off_bytes + (ch0 + 16 * s + 8 * h + 4 * q4) * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) r[8 * s + 4 * q4 + j] = v[j];
            }
        return r;
    };

    // ---- GEMM2 (k-outer) with the depthwise stage one k-step ahead --------------------------------------------
    floatx16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 3) * R) : "memory");     // slots 0 and 1 have landed
    __syncthreads();                                           // slots 0 and 1, slab 0, tables visible
    PS_STAMP(1);
    halo_commit(1);
    read_q(ic<0>{});
    read_q(ic<1>{});
    dw_load(0);
    half8 d_cur = dw_math();
    static_for<NT>([&](auto ii) {                              // slot i: k-step 2 i (groups 0, 1), 2 i + 1 (groups 2, 3)
        constexpr int i = decltype(ii)::value;
        if constexpr (i == 0) {                                // (the barrier above was step 0's)
            ring_issue_begin(NBUF - 1);
        } else {
            ring_step(ii, i);
            // slab s+1 is written at slot 2 s (it is first read one k-step before slot 2 s + 2) and requested at 2 s - 1
#ifdef DCVC_DIAG
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long th0 = __builtin_amdgcn_s_memtime();
#endif
            if constexpr ((i & 1) == 0 && i / 2 + 1 < CF::NSLAB) halo_commit(i / 2 + 1);
            if constexpr ((i & 1) == 1 && (i + 1) / 2 + 1 < CF::NSLAB) halo_load((i + 1) / 2 + 1);
#ifdef DCVC_DIAG
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            t_lds += __builtin_amdgcn_s_memtime() - th0;       // (GEMM2 phase: reported as "lds_drain" = halo staging)
#endif
        }
        if constexpr (i == (NT > 4 ? NT - 4 : 0)) {           // x' is needed right after this GEMM: request it 4 steps ahead
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                idf[ks] = VEC16_ZERO;
                if (pvalid) idf[ks] = *reinterpret_cast<const Vec16*>(ident + gpix * p.ldi + ks * 16 + h * 8);
            }
        }
        half8 d_n1 = d_cur, d_n2 = d_cur;
        slot_groups(ii, [&](auto jc, const half8 (&F)[GS]) {
            constexpr int j = decltype(jc)::value;
            if constexpr (j == 0) dw_load(2 * i + 1);                              // taps of the slot's second k-step
            if constexpr (j == 2) {
                d_cur = d_n1;
                dw_load(2 * i + 2 < KS ? 2 * i + 2 : KS - 1);                      // taps of the next slot's first k-step
            }
#pragma unroll
            for (int t = 0; t < GS; ++t) acc[(j & 1) * GS + t] = mfma32(F[t], d_cur, acc[(j & 1) * GS + t]);
            if constexpr (j == 0 || j == 2) __builtin_amdgcn_sched_barrier(0);     // reads issued here, used one group later
            if constexpr (j == 1) d_n1 = dw_math();
            if constexpr (j == 3) d_n2 = dw_math();
        });
        d_cur = d_n2;
    });

    PS_STAMP(2);
    // o = (W2 d + b2) + x', kept as the B fragments of the FFN (and as the residual of the block output)
    half8 o[KS];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const floatx16 b = bias16(CF::T_B2, 32 * t);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const half8 idh = __builtin_bit_cast(half8, idf[2 * t + s]);
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (acc[t][8 * s + j] + b[8 * s + j]) + (float)idh[j];
            o[2 * t + s] = pack8(v);
        }
    }

    // ---- FFN --------------------------------------------------------------------------------------------------
    // slots from G0 = NT on: lo(0) hi(0) | lo(1) hi(1) w4(0) | lo(2) hi(2) w4(1) | ... | w4(KS-1)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    // one 32-row u tile = one slot: FA holds its k-steps 0..NT-1 on entry, FB gets NT..KS-1, FA the next slot's first half
    auto u_slot = [&](auto gph, int g, floatx16& u, int bias_off) {
        ring_step(gph, g);
        u = bias16(CF::T_B3, bias_off);
        slot_groups(gph, [&](auto jc, const half8 (&F)[GS]) {
            constexpr int j = decltype(jc)::value;
#pragma unroll
            for (int k = 0; k < GS; ++k) u = mfma32(F[k], o[j * GS + k], u);
        });
    };
    // one W4 slot: v tile's two k-steps into the C/32 output tiles
    auto w4_slot = [&](auto gph, int g, const half8 (&vf)[2]) {
        ring_step(gph, g);
        slot_groups(gph, [&](auto jc, const half8 (&F)[GS]) {
            constexpr int j = decltype(jc)::value;
#pragma unroll
            for (int t = 0; t < GS; ++t) acc[(j & 1) * GS + t] = mfma32(F[t], vf[j >> 1], acc[(j & 1) * GS + t]);
        });
    };
    // step j: [tile j+1's u_lo slot | first half of tile j's gate] [u_hi slot | second half] [tile j's W4 slot]
    // G = phase of the step's first slot; g = its real index
    auto ffn_step = [&](auto gph, auto more_c, int g, int j, const floatx16& ulo, const floatx16& uhi, floatx16& nlo,
                        floatx16& nhi) {
        constexpr int G = decltype(gph)::value;
        constexpr bool more = decltype(more_c)::value != 0;
        float v[16];
        if constexpr (more) u_slot(ic<G>{}, g, nlo, 32 * (j + 1));
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = Traits<half_t>::gate2(ulo[i], uhi[i]);
        if constexpr (more) u_slot(ic<G + 1>{}, g + 1, nhi, 2 * C + 32 * (j + 1));
#pragma unroll
        for (int i = 8; i < 16; ++i) v[i] = Traits<half_t>::gate2(ulo[i], uhi[i]);
        half8 vf[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float w[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) w[i] = v[8 * s + i];
            vf[s] = pack8(w);
        }
        w4_slot(ic<G + (more ? 2 : 0)>{}, g + (more ? 2 : 0), vf);
    };
    PS_STAMP(3);
    constexpr int G0 = NT;                                     // phase bookkeeping: phases only matter mod NBUF
    floatx16 ua_lo, ua_hi, ub_lo, ub_hi;                       // two named tile pairs (no register-array indexing)
    u_slot(ic<G0>{}, G0, ua_lo, 0);
    u_slot(ic<G0 + 1>{}, G0 + 1, ua_hi, 2 * C);
    constexpr int GF = G0 + 2;                                 // first slot of step 0
    static_assert(12 % NBUF == 0 && KS % 4 == 0, "the FFN loop advances 4 steps = 12 slots per iteration");
    int g = GF;
    for (int j = 0; j + 4 < KS; j += 4, g += 12) {
        ffn_step(ic<GF + 0>{}, ic<1>{}, g + 0, j + 0, ua_lo, ua_hi, ub_lo, ub_hi);
        ffn_step(ic<GF + 3>{}, ic<1>{}, g + 3, j + 1, ub_lo, ub_hi, ua_lo, ua_hi);
        ffn_step(ic<GF + 6>{}, ic<1>{}, g + 6, j + 2, ua_lo, ua_hi, ub_lo, ub_hi);
        ffn_step(ic<GF + 9>{}, ic<1>{}, g + 9, j + 3, ub_lo, ub_hi, ua_lo, ua_hi);
    }
    ffn_step(ic<GF + 0>{}, ic<1>{}, g + 0, KS - 4, ua_lo, ua_hi, ub_lo, ub_hi);
    ffn_step(ic<GF + 3>{}, ic<1>{}, g + 3, KS - 3, ub_lo, ub_hi, ua_lo, ua_hi);
    ffn_step(ic<GF + 6>{}, ic<1>{}, g + 6, KS - 2, ua_lo, ua_hi, ub_lo, ub_hi);
    ffn_step(ic<GF + 9>{}, ic<0>{}, g + 9, KS - 1, ub_lo, ub_hi, ua_lo, ua_hi);
    g += 10;
    constexpr int GH = GF + 10;                                // phase of the first slot after the tail ( = TAIL_SLOTS mod 12 )

    PS_STAMP(4);
    // ---- r = (W4 v + b4) + o  (rounded to fp16 like every stored activation) -------------------------------------
    half8 r[KS];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const floatx16 b = bias16(CF::T_B4, 32 * t);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (acc[t][8 * s + j] + b[8 * s + j]) + (float)o[2 * t + s][j];
            r[2 * t + s] = pack8(v);
        }
    }

    // ---- output: staged through LDS 64 channels at a time, written with full 128-byte lines -------------------
    // staging buffer: [128 px][64 ch] halves, pixel row = 128 bytes, chunk c of pixel n in slot c ^ (n & 7)
    char* stage = smem + CF::STAGE_OFF;
    const int lp = wave * 32 + pp;                             // pixel index inside the tile (row-major 8 x 16)
    half_t* out = reinterpret_cast<half_t*>(p.out);
    auto store_tile = [&](const half8 (&fr)[KS], half_t* dst, long ld, bool epilogue) {
#pragma unroll
        for (int sl = 0; sl < CF::NSLAB; ++sl) {                // (unrolled: the fragment index must be static)
            char* sb = stage + (sl & 1) * 16384;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int c = 2 * kk + h;
                *reinterpret_cast<half8*>(sb + lp * 128 + ((c ^ (lp & 7)) << 4)) = fr[sl * 4 + kk];
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int it = tid + k * NTHR;
                const int n = it >> 3, c = it & 7;
                const int y = ty0 + (n >> 4), x = tx0 + (n & 15);
                if (y < p.H && x < p.W) {
                    const long pix = (long)y * p.W + x;
                    const int ch = sl * SLAB + c * 8;
                    Vec16 raw = *reinterpret_cast<const Vec16*>(sb + n * 128 + ((c ^ (n & 7)) << 4));
                    if (epilogue && (p.shortcut || p.q != nullptr)) {
                        float f[8];
                        unpack16<half_t>(raw, f);
                        if (p.shortcut) {
                            float id[8];
                            unpack16<half_t>(*reinterpret_cast<const Vec16*>(ident + pix * p.ldi + ch), id);
#pragma unroll
                            for (int j = 0; j < 8; ++j) f[j] = f[j] + id[j];
                        }
                        if (p.q != nullptr) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) f[j] = f[j] * ((ch + j) < p.c_log ? p.q[ch + j] : 1.0f);
                        }
                        raw = pack16<half_t>(f);
                    }
                    *reinterpret_cast<Vec16*>(dst + pix * ld + ch) = raw;
                }
            }
            // (two staging buffers: the next slab's writes go to the other one; its barrier orders this slab's reads
            //  against the writes two slabs later)
        }
    };
    __syncthreads();                                           // every wave is done with the halo region (depthwise)
    PS_STAMP(5);
    store_tile(r, out, p.ldo, true);
    PS_STAMP(6);

    // ---- fused head of the next block: a' = g(W1' r + b1') on the tile in registers (tile-outer) ------------------
    if (p.nstream != nullptr) {
        half8 an[KS];
        static_for<NT>([&](auto tt) {
            constexpr int t = decltype(tt)::value;
            floatx16 u;
            {
                constexpr int G = GH + t;
                ring_step(ic<G>{}, g + t);
                u = bias16(CF::T_NB1, 32 * t);
                slot_groups(ic<G>{}, [&](auto jc, const half8 (&F)[GS]) {
                    constexpr int j = decltype(jc)::value;
#pragma unroll
                    for (int k = 0; k < GS; ++k) u = mfma32(F[k], r[j * GS + k], u);
                });
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = gate1(u[8 * s + j]);
                an[2 * t + s] = pack8(v);
            }
        });
        __syncthreads();
        store_tile(an, reinterpret_cast<half_t*>(p.na_out), p.nlda, false);
    }
    PS_STAMP(7);
#ifdef DCVC_DIAG
    if (p.stamps && (threadIdx.x & 63) == 0) {      // per wave: cycles waited in ring_step (LDS drain | barrier | commit)
        unsigned long long* w = p.stamps + (size_t)gridDim.x * 8 + ((size_t)blockIdx.x * 4 + wave) * 3;
        w[0] = t_lds;
        w[1] = t_bar;
        w[2] = t_commit;
    }
#endif
}

}  // namespace ps
