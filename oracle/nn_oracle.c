/* nn_oracle.c - TEST INFRASTRUCTURE ONLY (parity checker; never linked into the product).
 *
 * Plain-C fp32 restatement of the DCVC-RT neural operators on the hot path.  Tensors are
 * row-major [pixel][channel] ("HWC"); the numpy front-end (dcvc_oracle.py) converts from/to
 * the reference's NCHW at the boundary.  The order of every floating-point operation is
 * fixed (k-ascending fused-multiply-add chains, bias added after the chain, residuals after
 * the bias) and is the order the fp32 "exact" HIP kernels use, so the HIP fp32 path can be
 * compared bit-for-bit with this file; against the reference (torch CPU) it agrees to fp32
 * rounding (pinned by tests/golden fixtures generated from the reference itself).
 *
 * Reference functions restated (all paths relative to /root/reference):
 *   nn.Conv2d 1x1 / kxk (layers.py:70-81,138; video_model.py:57,63,103,109,141,199,212)
 *   depthwise 3x3, zero pad  (layers.py:75)
 *   WSiLU / WSiLUChunkAdd    (layers.py:11-26)
 *   DepthConvBlock.forward_torch (layers.py:92-106)
 *   PixelShuffle / pixel_unshuffle (layers.py:34; video_model.py:68,153)
 *   elementwise entropy glue (cuda_inference.py:26-171 else-branches)
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "dcvc_math.h"

#define PB 4 /* pixel block */

/* out[p][n] = (sum_k x[p][k] * w[n][k]) + b[n]      w: torch layout [N][K]
 * x rows have stride ldx, out rows stride ldo.  b may be NULL. */
void orc_conv1x1(const float* x, int64_t ldx, const float* w, const float* b, float* out,
                 int64_t ldo, int64_t P, int K, int N)
{
    float* wt = (float*)malloc(sizeof(float) * (size_t)K * N);
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) wt[(size_t)k * N + n] = w[(size_t)n * K + k];
#pragma omp parallel
    {
        float* acc = (float*)malloc(sizeof(float) * PB * (size_t)N);
#pragma omp for schedule(static)
        for (int64_t p0 = 0; p0 < P; p0 += PB) {
            const int np = (int)((P - p0) < PB ? (P - p0) : PB);
            memset(acc, 0, sizeof(float) * PB * (size_t)N);
            for (int k = 0; k < K; ++k) {
                const float* wr = wt + (size_t)k * N;
                for (int j = 0; j < np; ++j) {
                    const float xv = x[(p0 + j) * ldx + k];
                    float* a = acc + (size_t)j * N;
                    for (int n = 0; n < N; ++n) a[n] = __builtin_fmaf(xv, wr[n], a[n]);
                }
            }
            for (int j = 0; j < np; ++j) {
                float* o = out + (p0 + j) * ldo;
                const float* a = acc + (size_t)j * N;
                if (b)
                    for (int n = 0; n < N; ++n) o[n] = a[n] + b[n];
                else
                    for (int n = 0; n < N; ++n) o[n] = a[n];
            }
        }
        free(acc);
    }
    free(wt);
}

/* General dense conv, HWC.  w: torch layout [N][Cin][KH][KW].
 * K-order of the fma chain: (ky, kx) major, cin minor; out-of-bounds taps are skipped. */
void orc_conv2d(const float* x, int H, int W, int Cin, const float* w, const float* b, float* out,
                int N, int KH, int KW, int stride, int pad)
{
    const int Ho = (H + 2 * pad - KH) / stride + 1;
    const int Wo = (W + 2 * pad - KW) / stride + 1;
    /* repack to [tap][cin][n] */
    float* wt = (float*)malloc(sizeof(float) * (size_t)KH * KW * Cin * N);
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < Cin; ++c)
            for (int t = 0; t < KH * KW; ++t)
                wt[((size_t)t * Cin + c) * N + n] = w[((size_t)n * Cin + c) * KH * KW + t];
#pragma omp parallel
    {
        float* acc = (float*)malloc(sizeof(float) * (size_t)N);
#pragma omp for schedule(static) collapse(2)
        for (int oy = 0; oy < Ho; ++oy)
            for (int ox = 0; ox < Wo; ++ox) {
                memset(acc, 0, sizeof(float) * (size_t)N);
                for (int ky = 0; ky < KH; ++ky) {
                    const int iy = oy * stride - pad + ky;
                    if (iy < 0 || iy >= H) continue;
                    for (int kx = 0; kx < KW; ++kx) {
                        const int ix = ox * stride - pad + kx;
                        if (ix < 0 || ix >= W) continue;
                        const float* xr = x + ((size_t)iy * W + ix) * Cin;
                        const float* wr = wt + (size_t)(ky * KW + kx) * Cin * N;
                        for (int c = 0; c < Cin; ++c) {
                            const float xv = xr[c];
                            const float* wrr = wr + (size_t)c * N;
                            for (int n = 0; n < N; ++n) acc[n] = __builtin_fmaf(xv, wrr[n], acc[n]);
                        }
                    }
                }
                float* o = out + ((size_t)oy * Wo + ox) * N;
                if (b)
                    for (int n = 0; n < N; ++n) o[n] = acc[n] + b[n];
                else
                    for (int n = 0; n < N; ++n) o[n] = acc[n];
            }
        free(acc);
    }
    free(wt);
}

/* depthwise 3x3, zero padding 1.  w: torch layout [C][1][3][3]; taps in (ky,kx) order, OOB skipped,
 * bias added after the chain. */
void orc_dw3x3(const float* x, int H, int W, int C, const float* w, const float* b, float* out)
{
    float* wt = (float*)malloc(sizeof(float) * 9 * (size_t)C);
    for (int c = 0; c < C; ++c)
        for (int t = 0; t < 9; ++t) wt[(size_t)t * C + c] = w[(size_t)c * 9 + t];
#pragma omp parallel for schedule(static) collapse(2)
    for (int y = 0; y < H; ++y)
        for (int xx = 0; xx < W; ++xx) {
            float* o = out + ((size_t)y * W + xx) * C;
            for (int c = 0; c < C; ++c) o[c] = 0.0f;
            for (int ky = 0; ky < 3; ++ky) {
                const int iy = y + ky - 1;
                if (iy < 0 || iy >= H) continue;
                for (int kx = 0; kx < 3; ++kx) {
                    const int ix = xx + kx - 1;
                    if (ix < 0 || ix >= W) continue;
                    const float* xr = x + ((size_t)iy * W + ix) * C;
                    const float* wr = wt + (size_t)(ky * 3 + kx) * C;
                    for (int c = 0; c < C; ++c) o[c] = __builtin_fmaf(xr[c], wr[c], o[c]);
                }
            }
            if (b)
                for (int c = 0; c < C; ++c) o[c] = o[c] + b[c];
        }
    free(wt);
}

void orc_wsilu(const float* x, float* out, int64_t n)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) out[i] = dcvc_wsiluf(x[i]);
}

/* out[p][c] = wsilu(x[p][c]) + wsilu(x[p][c + C2]),  x: [P][2*C2]   (layers.py:19-26) */
void orc_wsilu_chunk_add(const float* x, float* out, int64_t P, int C2)
{
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < P; ++p) {
        const float* xr = x + p * 2 * C2;
        float* o = out + p * C2;
        for (int c = 0; c < C2; ++c) o[c] = dcvc_wsiluf(xr[c]) + dcvc_wsiluf(xr[c + C2]);
    }
}

void orc_sigmoid(const float* x, float* out, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) out[i] = dcvc_sigmoidf(x[i]);
}

void orc_round(const float* x, float* out, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) out[i] = dcvc_roundf(x[i]);
}

void orc_scale_to_index(const float* s, uint8_t* out, int64_t n, float smin, float smax,
                        float log_smin, float log_step_recip)
{
    for (int64_t i = 0; i < n; ++i)
        out[i] = dcvc_scale_to_index(s[i], smin, smax, log_smin, log_step_recip);
}

/* DepthConvBlock.forward_torch (layers.py:92-106), HWC.
 *   x' = adaptor(x)                       (if wa != NULL)
 *   o  = conv2(dw(wsilu(conv1(x')))) + x'
 *   r  = ffn2(chunk_add(wsilu(ffn1(o)))) + o   (+ x' if shortcut) (* q[c] if q != NULL)
 * out rows have stride ldo (lets the caller place the result inside a channel-concat buffer). */
void orc_dcb(const float* x, int64_t ldx, int H, int W, int Cin, int C, const float* wa,
             const float* ba, const float* w1, const float* b1, const float* wd, const float* bd,
             const float* w2, const float* b2, const float* w3, const float* b3, const float* w4,
             const float* b4, int shortcut, const float* q, float* out, int64_t ldo)
{
    const int64_t P = (int64_t)H * W;
    float* xi = (float*)malloc(sizeof(float) * P * C);
    float* t = (float*)malloc(sizeof(float) * P * C);
    float* d = (float*)malloc(sizeof(float) * P * C);
    float* o = (float*)malloc(sizeof(float) * P * C);
    float* u = (float*)malloc(sizeof(float) * P * 4 * C);
    float* v = (float*)malloc(sizeof(float) * P * 2 * C);
    if (wa) {
        orc_conv1x1(x, ldx, wa, ba, xi, C, P, Cin, C);
    } else {
        for (int64_t p = 0; p < P; ++p) memcpy(xi + p * C, x + p * ldx, sizeof(float) * C);
    }
    orc_conv1x1(xi, C, w1, b1, t, C, P, C, C);
    orc_wsilu(t, t, P * C);
    orc_dw3x3(t, H, W, C, wd, bd, d);
    orc_conv1x1(d, C, w2, b2, o, C, P, C, C);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < P * C; ++i) o[i] = o[i] + xi[i];
    orc_conv1x1(o, C, w3, b3, u, 4 * C, P, C, 4 * C);
    orc_wsilu_chunk_add(u, v, P, 2 * C);
    orc_conv1x1(v, 2 * C, w4, b4, t, C, P, 2 * C, C);
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < P; ++p) {
        float* orow = out + p * ldo;
        for (int c = 0; c < C; ++c) {
            float r = t[p * C + c] + o[p * C + c];
            if (shortcut) r = r + xi[p * C + c];
            if (q) r = r * q[c];
            orow[c] = r;
        }
    }
    free(xi); free(t); free(d); free(o); free(u); free(v);
}
