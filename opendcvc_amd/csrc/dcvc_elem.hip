// dcvc_elem.hip - HBM-bound elementwise / layout / entropy-glue kernels of the DCVC-RT path.
//  (1) pipeline forms on HWC tensors with the checkerboard masks computed from (h, w, c);
//  (2) the flat NCHW forms of the reference's operator module (inference_extensions_cuda,
//      kernel.cu:56-1004) with the reference's own signatures, for the operator seam.
// All arithmetic is fp32 with the shared deterministic math of include/dcvc_math.h; storage is
// _Float16 or float.
#include "common.hpp"
#include "gemm_core.hpp"

namespace {

constexpr int EB = 256;   // threads per block for 1-D kernels

inline int nblocks(int64_t n) { return (int)((n + EB - 1) / EB); }

template <typename T>
__device__ __forceinline__ float ld(const T* p, int64_t i)
{
    return (float)p[i];
}
// fp32 VALUE -> storage type.  The empty asm hides the value's producer, so the compiler cannot fold the preceding
// fp32 add / multiply into v_fma_mixlo_f16 (one rounding) in one kernel and leave add + cvt (two roundings) in another:
// the encoder's and the decoder's y_hat kernels must round alike (see Traits<half_t>::from_f, gemm_core.hpp).
template <typename T>
__device__ __forceinline__ T to_t(float v)
{
    asm("" : "+v"(v));
    return (T)v;
}
template <typename T>
__device__ __forceinline__ void st(T* p, int64_t i, float v)
{
    p[i] = to_t<T>(v);
}

__device__ __forceinline__ float clampf(float v, float lo, float hi)
{
    v = v < lo ? lo : v;
    return v > hi ? hi : v;
}

constexpr float kScaleMin = 0.11f, kScaleMax = 16.0f;
// log(0.11) and 127 / (log(16) - log(0.11)) rounded to float exactly as the reference's python
// floats are when passed into torch fp32 expressions (entropy_models.py:235-238)
constexpr float kLogScaleMin = -2.2072749131897207f;
constexpr float kLogStepRecip = 25.50270635855404f;

// active channel group for checkerboard step `step` at pixel (h, w): see common_model.py:99-131
__device__ __forceinline__ int active_group(int n_groups, int step, int h, int w)
{
    if (n_groups == 2) return ((h + w) & 1) ^ (step & 1);
    const int pos = ((h & 1) << 1) | (w & 1);
    const int x = (step == 0) ? 0 : (step == 1) ? 3 : (step == 2) ? 2 : 1;
    return pos ^ x;
}

// ------------------------------------------------------------------ layout kernels
template <typename T>
__global__ void unshuffle8_kernel(const T* x, int C, int H, int W, T* out, int64_t ldo)
{
    const int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x;
    if (i >= (int64_t)C * H * W) return;
    const int xw = (int)(i % W), y = (int)((i / W) % H), c = (int)(i / ((int64_t)W * H));
    const int W8 = W / 8;
    out[((int64_t)(y >> 3) * W8 + (xw >> 3)) * ldo + c * 64 + (y & 7) * 8 + (xw & 7)] = x[i];
}

// PixelUnshuffle(8) / PixelShuffle(8) through LDS: a block moves 32 output pixels of one row of the HWC map
// = C x 8 image rows x 256 columns.  Image rows are read / written as contiguous 512-byte (f16) runs,
// the HWC side as one contiguous run of 32 pixels x 64C channels; both sides use 16-byte accesses.
constexpr int S8_PIX = 32;

template <typename T>
__global__ __launch_bounds__(EB) void unshuffle8_tiled_kernel(const T* x, int C, int H, int W, T* out, int64_t ldo)
{
    constexpr int V = Traits<T>::kVec, COLS = S8_PIX * 8, VPR = COLS / V;   // vectors per image-row segment
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* tile = reinterpret_cast<T*>(smem);                                   // [C*8][COLS]
    const int W8 = W / 8, segs = (W8 + S8_PIX - 1) / S8_PIX;
    const int oh = blockIdx.x / segs, ow0 = (blockIdx.x % segs) * S8_PIX;
    const int npix = min(S8_PIX, W8 - ow0);
    const int rows = C * 8;
    for (int it = threadIdx.x; it < rows * VPR; it += EB) {
        const int r = it / VPR, v = it - r * VPR;                          // r = c*8 + (y & 7)
        if (v * V < npix * 8) {
            const int c = r >> 3, y = oh * 8 + (r & 7);
            *reinterpret_cast<Vec16*>(tile + r * COLS + v * V) =
                *reinterpret_cast<const Vec16*>(x + ((int64_t)c * H + y) * W + ow0 * 8 + v * V);
        }
    }
    __syncthreads();
    constexpr int G = 8 / V;                                               // 16-byte vectors per 8-element group
    for (int it = threadIdx.x; it < npix * rows * G; it += EB) {
        const int pix = it / (rows * G), rem = it - pix * (rows * G), r = rem / G, g = rem - r * G;
        *reinterpret_cast<Vec16*>(out + ((int64_t)oh * W8 + ow0 + pix) * ldo + r * 8 + g * V) =
            *reinterpret_cast<const Vec16*>(tile + r * COLS + pix * 8 + g * V);
    }
}

template <typename T>
__global__ __launch_bounds__(EB) void shuffle8_tiled_kernel(const T* x, int64_t ldx, const float* bias, int C, int H, int W,
                                                           int do_clamp, T* out)
{
    constexpr int V = Traits<T>::kVec, COLS = S8_PIX * 8, VPR = COLS / V, G = 8 / V;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* tile = reinterpret_cast<T*>(smem);                                   // [C*8][COLS]
    const int segs = (W + S8_PIX - 1) / S8_PIX;
    const int oh = blockIdx.x / segs, ow0 = (blockIdx.x % segs) * S8_PIX;
    const int npix = min(S8_PIX, W - ow0);
    const int rows = C * 8, HO = H * 8, WO = W * 8;
    for (int it = threadIdx.x; it < npix * rows * G; it += EB) {
        const int pix = it / (rows * G), rem = it - pix * (rows * G), r = rem / G, g = rem - r * G;
        float v[V];
        unpack16<T>(*reinterpret_cast<const Vec16*>(x + ((int64_t)oh * W + ow0 + pix) * ldx + r * 8 + g * V), v);
#pragma unroll
        for (int j = 0; j < V; ++j) {
            if (bias) v[j] = v[j] + bias[r * 8 + g * V + j];
            if (do_clamp) v[j] = clampf(v[j], 0.f, 1.f);
        }
        *reinterpret_cast<Vec16*>(tile + r * COLS + pix * 8 + g * V) = pack16<T>(v);
    }
    __syncthreads();
    for (int it = threadIdx.x; it < rows * VPR; it += EB) {
        const int r = it / VPR, v = it - r * VPR;
        if (v * V < npix * 8) {
            const int c = r >> 3, y = oh * 8 + (r & 7);
            *reinterpret_cast<Vec16*>(out + ((int64_t)c * HO + y) * WO + ow0 * 8 + v * V) =
                *reinterpret_cast<const Vec16*>(tile + r * COLS + v * V);
        }
    }
}

template <typename T>
__global__ void shuffle8_kernel(const T* x, int64_t ldx, const float* bias, int C, int H, int W, int do_clamp, T* out)
{
    const int HO = H * 8, WO = W * 8;
    const int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x;
    if (i >= (int64_t)C * HO * WO) return;
    const int xw = (int)(i % WO), y = (int)((i / WO) % HO), c = (int)(i / ((int64_t)WO * HO));
    const int ch = c * 64 + (y & 7) * 8 + (xw & 7);
    float v = ld(x, ((int64_t)(y >> 3) * W + (xw >> 3)) * ldx + ch);
    if (bias) v = v + bias[ch];
    if (do_clamp) v = clampf(v, 0.f, 1.f);
    st(out, i, v);
}

template <typename T>
__global__ void replicate_pad_hwc_kernel(const T* x, int64_t ldx, int H, int W, int C, int HO, int WO, T* out, int64_t ldo)
{
    const int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x;
    if (i >= (int64_t)HO * WO * C) return;
    const int c = (int)(i % C);
    const int64_t p = i / C;
    const int xw = (int)(p % WO), y = (int)(p / WO);
    const int sy = y < H ? y : H - 1, sx = xw < W ? xw : W - 1;
    out[p * ldo + c] = x[((int64_t)sy * W + sx) * ldx + c];
}

template <typename T>
__global__ void scale_channels_kernel(const T* x, int64_t ldx, const float* q, int64_t P, int C, T* out, int64_t ldo)
{
    const int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x;
    if (i >= P * C) return;
    const int c = (int)(i % C);
    const int64_t p = i / C;
    st(out, p * ldo + c, ld(x, p * ldx + c) * q[c]);
}

// same, 16 bytes per thread (C, ldx, ldo multiples of the vector width, 16-byte aligned bases)
template <typename T>
__global__ void scale_channels_vec_kernel(const T* x, int64_t ldx, const float* q, int64_t P, int C, T* out, int64_t ldo)
{
    constexpr int V = Traits<T>::kVec;
    const int gc = C / V;
    const int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x;
    if (i >= P * gc) return;
    const int c = (int)(i % gc) * V;
    const int64_t p = i / gc;
    float v[V];
    unpack16<T>(*reinterpret_cast<const Vec16*>(x + p * ldx + c), v);
#pragma unroll
    for (int j = 0; j < V; ++j) v[j] = v[j] * q[c + j];
    *reinterpret_cast<Vec16*>(out + p * ldo + c) = pack16<T>(v);
}

template <typename T>
__global__ void copy_channels_kernel(const T* x, int64_t ldx, int64_t P, int C, T* out, int64_t ldo)
{
    const int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x;
    if (i >= P * C) return;
    const int c = (int)(i % C);
    const int64_t p = i / C;
    out[p * ldo + c] = x[p * ldx + c];
}

template <typename T>
__global__ void crop_hwc_kernel(const T* x, int64_t ldx, int W, int H2, int W2, int C, T* out, int64_t ldo)
{
    const int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x;
    if (i >= (int64_t)H2 * W2 * C) return;
    const int c = (int)(i % C);
    const int64_t p = i / C;
    const int xw = (int)(p % W2), y = (int)(p / W2);
    out[p * ldo + c] = x[((int64_t)y * W + xw) * ldx + c];
}

template <typename T>
__global__ void nchw_to_hwc_kernel(const T* x, int C, int64_t HW, T* out, int64_t ldo)
{
    const int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x;
    if (i >= HW * C) return;
    const int c = (int)(i % C);
    const int64_t p = i / C;
    out[p * ldo + c] = x[(int64_t)c * HW + p];
}

template <typename T>
__global__ void hwc_to_nchw_kernel(const T* x, int64_t ldx, int C, int64_t HW, T* out)
{
    const int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x;
    if (i >= HW * C) return;
    const int64_t p = i % HW;
    const int c = (int)(i / HW);
    out[i] = x[p * ldx + c];
}


// ------------------------------------------------------------------ YUV 4:2:0 <-> model frame
template <typename T>
__global__ void yuv420_to_frame_kernel(const uint8_t* yp, const uint8_t* up, const uint8_t* vp, int H, int W, int HO,
                                       int WO, T* out)
{
    const int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x;
    if (i >= (int64_t)3 * HO * WO) return;
    const int xw = (int)(i % WO), y = (int)((i / WO) % HO), c = (int)(i / ((int64_t)WO * HO));
    const int sy = y < H ? y : H - 1, sx = xw < W ? xw : W - 1;        // replicate pad of the 4:4:4 frame
    int v;
    if (c == 0)
        v = yp[(int64_t)sy * W + sx];
    else
        v = (c == 1 ? up : vp)[(int64_t)(sy >> 1) * (W >> 1) + (sx >> 1)];   // nearest chroma upsampling
    st(out, i, (float)v / 255.0f);
}

template <typename T>
__global__ void frame_to_yuv420_kernel(const T* x, int HP, int WP, int H, int W, int round_uv, uint8_t* yp, uint8_t* up,
                                       uint8_t* vp)
{
    const int64_t ny = (int64_t)H * W, nc = (int64_t)(H >> 1) * (W >> 1);
    const int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x;
    if (i >= ny + 2 * nc) return;
    if (i < ny) {
        const int xw = (int)(i % W), y = (int)(i / W);
        const float s = (float)to_t<T>(ld(x, (int64_t)y * WP + xw) * 255.0f);   // product rounded to the storage type, as torch does
        yp[i] = (uint8_t)dcvc_roundf(clampf(s, 0.f, 255.f));
        return;
    }
    const int64_t j = i - ny;
    const int c = j < nc ? 1 : 2;
    const int64_t k = c == 1 ? j : j - nc;
    const int w2 = W >> 1;
    const int xw = (int)(k % w2), y = (int)(k / w2);
    const T* pl = x + (int64_t)c * HP * WP;
    const float a = ld(pl, (int64_t)(2 * y) * WP + 2 * xw), b = ld(pl, (int64_t)(2 * y) * WP + 2 * xw + 1);
    const float d = ld(pl, (int64_t)(2 * y + 1) * WP + 2 * xw), e = ld(pl, (int64_t)(2 * y + 1) * WP + 2 * xw + 1);
    const float m = (float)to_t<T>(((a + b) + (d + e)) * 0.25f);                 // avg_pool2d(2) result in the storage type
    float s = clampf((float)to_t<T>(m * 255.0f), 0.f, 255.f);
    if (round_uv) s = dcvc_roundf(s);
    (c == 1 ? up : vp)[k] = (uint8_t)s;                                      // truncation like `.to(uint8)`
}

// ------------------------------------------------------------------ RGB (PNG sources) <-> model frame
// ITU-R BT.709 weights as the reference's transforms.py:7-10 spells them; the scalars below are the fp32 values torch
// uses when a Python float meets a float32 / float16 tensor.
__device__ __forceinline__ void rgb_consts(float& kr, float& kg, float& kb)
{
    kr = 0.2126f;
    kg = 0.7152f;
    kb = 0.0722f;
}

// uint8 planar RGB [3][H][W] -> padded YCbCr model input: /255 (test_video.py:60-63), rgb2ycbcr in fp32 with the reference's
// operation order and clamp (transforms.py:27-38), ONE rounding to the storage type (test_video.py:90), replicate pad (:179)
template <typename T>
__global__ void rgb_to_frame_kernel(const uint8_t* rgb, int H, int W, int HO, int WO, T* out)
{
    const int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x;
    if (i >= (int64_t)HO * WO) return;
    const int xw = (int)(i % WO), y = (int)(i / WO);
    const int sy = y < H ? y : H - 1, sx = xw < W ? xw : W - 1;
    const int64_t sp = (int64_t)sy * W + sx, plane = (int64_t)H * W;
    const float r = (float)rgb[sp] / 255.0f, g = (float)rgb[plane + sp] / 255.0f, b = (float)rgb[2 * plane + sp] / 255.0f;
    float kr, kg, kb;
    rgb_consts(kr, kg, kb);
    const float yy = (kr * r + kg * g) + kb * b;
    const float cb = (0.5f * (b - yy)) / (float)(1.0 - 0.0722) + 0.5f;
    const float cr = (0.5f * (r - yy)) / (float)(1.0 - 0.2126) + 0.5f;
    const int64_t op = (int64_t)HO * WO;
    st(out, i, clampf(yy, 0.f, 1.f));
    st(out, op + i, clampf(cb, 0.f, 1.f));
    st(out, 2 * op + i, clampf(cr, 0.f, 1.f));
}

// reconstruction [3][HP][WP] (YCbCr) -> clamp(ycbcr2rgb(x) * 255, 0, 255) of the HxW picture as [3][H][W] in the storage type:
// every operation of transforms.py:41-53 + test_video.py:118-119 rounded to the storage type like the reference's tensors
// (fp16 reconstructions are converted in fp16 there); what the reference's RGB PSNR / MS-SSIM / PNG writer read
template <typename T>
__global__ void frame_to_rgb_kernel(const T* x, int HP, int WP, int H, int W, T* out)
{
    const int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x;
    if (i >= (int64_t)H * W) return;
    const int xw = (int)(i % W), y = (int)(i / W);
    const int64_t sp = (int64_t)y * WP + xw, plane = (int64_t)HP * WP;
    auto q = [](float v) { return (float)to_t<T>(v); };
    const float yy = ld(x, sp), cb = ld(x, plane + sp), cr = ld(x, 2 * plane + sp);
    float kr, kg, kb;
    rgb_consts(kr, kg, kb);
    // (a Python scalar next to a half tensor enters torch's kernels in fp32 - the op-math type - not rounded to fp16)
    const float sr = (float)(2.0 - 2.0 * 0.2126), sb = (float)(2.0 - 2.0 * 0.0722);
    const float r = q(yy + q(sr * q(cr - 0.5f)));
    const float b = q(yy + q(sb * q(cb - 0.5f)));
    const float g = q(q(q(yy - q(kr * r)) - q(kb * b)) / kg);
    const int64_t op = (int64_t)H * W;
    st(out, i, clampf(q(clampf(r, 0.f, 1.f) * 255.0f), 0.f, 255.f));
    st(out, op + i, clampf(q(clampf(g, 0.f, 1.f) * 255.0f), 0.f, 255.f));
    st(out, 2 * op + i, clampf(q(clampf(b, 0.f, 1.f) * 255.0f), 0.f, 255.f));
}

// ------------------------------------------------------------------ z quantiser
template <typename T>
__global__ void round_z_kernel(T* z, int64_t ldz, int64_t HW, int C, int8_t* z_chw)
{
    const int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x;
    if (i >= HW * C) return;
    const int c = (int)(i % C);
    const int64_t p = i / C;
    const float v = clampf(dcvc_roundf(ld(z, p * ldz + c)), -128.f, 127.f);
    st(z, p * ldz + c, v);
    z_chw[(int64_t)c * HW + p] = (int8_t)v;
}

template <typename T>
__global__ void z_from_int8_kernel(const int8_t* z_chw, int64_t HW, int C, T* out, int64_t ldo)
{
    // consecutive threads read consecutive bytes of the CHW symbol array: it may live in pinned HOST memory (the decoder
    // hands the coder's output over without a copy command), where a strided byte read is a bus transaction each; the
    // scattered 2 / 4-byte writes go to device memory (65 K elements at 1080p)
    const int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x;
    if (i >= HW * C) return;
    const int c = (int)(i / HW);
    const int64_t p = i - (int64_t)c * HW;
    st(out, p * ldo + c, (float)z_chw[i]);
}

// ------------------------------------------------------------------ checkerboard prior loop
// One block = 64 consecutive pixels x all collapsed channels.  Phase 1 walks (pixel, channel) with
// the channel fastest (coalesced HWC reads / y_hat writes); the packed symbols go through LDS so
// phase 2 can write them pixel-fastest in the reference's CHW order.
constexpr int PT = 16;

struct PriorEncArgs {
    int n_groups, step, q_mode;
    const void* y;
    int64_t ldy;
    const void* qsrc;
    int64_t ldq;
    const void* scales;
    int64_t lds;
    const void* means;
    int64_t ldm;
    int H, W, C;
    float thres;
    const void* yhat_in;
    int64_t ldhi;
    void* yhat_out;
    int64_t ldho;
    int16_t* packed;
};

template <typename T>
__global__ __launch_bounds__(EB) void prior_enc_kernel(PriorEncArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int16_t* sp = reinterpret_cast<int16_t*>(smem);   // [Cg][PT]
    const int Cg = a.C / a.n_groups;
    const int64_t HW = (int64_t)a.H * a.W;
    const int64_t p0 = (int64_t)blockIdx.x * PT;
    const T* y = (const T*)a.y;
    const T* qs = (const T*)a.qsrc;
    const T* sc = (const T*)a.scales;
    const T* mu = (const T*)a.means;
    const T* hin = (const T*)a.yhat_in;
    T* hout = (T*)a.yhat_out;
    for (int it = threadIdx.x; it < PT * Cg; it += EB) {
        const int pl = it / Cg, cc = it - pl * Cg;
        const int64_t p = p0 + pl;
        if (p >= HW) continue;
        const int h = (int)p / a.W, w = (int)p - h * a.W;      // (H * W < 2^31: 32-bit division)
        const int ga = active_group(a.n_groups, a.step, h, w);
        int16_t packed = 0;
        for (int g = 0; g < a.n_groups; ++g) {
            const int ch = cc + g * Cg;
            const float prev = a.step == 0 ? 0.f : ld(hin, p * a.ldhi + ch);
            if (g != ga) {
                st(hout, p * a.ldho + ch, prev);
                continue;
            }
            float qe;
            if (a.q_mode == 0) {
                float qd = ld(qs, p * a.ldq + ch);
                qd = qd < 0.5f ? 0.5f : qd;
                qe = 1.0f / qd;
            } else {
                qe = dcvc_sigmoidf(ld(qs, p * a.ldq)) * 1.5f + 0.5f;
            }
            const float yq = ld(y, p * a.ldy + ch) * qe;
            const float s = ld(sc, p * a.lds + ch), m = ld(mu, p * a.ldm + ch);
            float v = dcvc_roundf(yq - m);
            const bool use_thres = a.thres >= 0.f;
            if (use_thres && !(s > a.thres)) v = v * 0.f;
            v = clampf(v, -128.f, 127.f);
            // T-rounded like the reference, whose y_hat_k tensors are stored in the model dtype
            const float yh = (float)to_t<T>(v + m);
            st(hout, p * a.ldho + ch, a.step == 0 ? yh : prev + yh);
            const float scl = clampf(s, kScaleMin, kScaleMax);
            const bool keep = !use_thres || (scl > a.thres);
            const int idx = keep ? (int)dcvc_scale_to_index(scl, kScaleMin, kScaleMax, kLogScaleMin, kLogStepRecip) : 0xFF;
            packed = (int16_t)((int)v * 256 + idx);
        }
        sp[cc * PT + pl] = packed;
    }
    __syncthreads();
    for (int it = threadIdx.x; it < PT * Cg; it += EB) {
        const int cc = it / PT, pl = it - cc * PT;
        const int64_t p = p0 + pl;
        if (p < HW) a.packed[(int64_t)cc * HW + p] = sp[cc * PT + pl];
    }
}

template <typename T>
__global__ __launch_bounds__(EB) void prior_dec_index_kernel(int n_groups, int step, const T* sc, int64_t lds, int H, int W,
                                                            int C, float thres, uint8_t* idx_chw, uint8_t* cnt16)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint8_t* sp = reinterpret_cast<uint8_t*>(smem);   // [Cg][PT]
    const int Cg = C / n_groups;
    const int64_t HW = (int64_t)H * W;
    const int64_t p0 = (int64_t)blockIdx.x * PT;
    for (int it = threadIdx.x; it < PT * Cg; it += EB) {
        const int pl = it / Cg, cc = it - pl * Cg;
        const int64_t p = p0 + pl;
        if (p >= HW) continue;
        const int h = (int)p / W, w = (int)p - h * W;
        const int ch = cc + active_group(n_groups, step, h, w) * Cg;
        const float scl = clampf(ld(sc, p * lds + ch), kScaleMin, kScaleMax);
        const bool keep = thres < 0.f || scl > thres;
        sp[cc * PT + pl] = keep ? dcvc_scale_to_index(scl, kScaleMin, kScaleMax, kLogScaleMin, kLogStepRecip) : (uint8_t)0xFF;
    }
    __syncthreads();
    // kept entries of each (channel, 16-pixel run): what dec_compact_kernel scans (the compacted hand-off)
    if (cnt16 != nullptr) {
        for (int cc = threadIdx.x; cc < Cg; cc += EB) {
            int c = 0;
            for (int pl = 0; pl < PT; ++pl) c += (p0 + pl < HW && sp[cc * PT + pl] != 0xFF) ? 1 : 0;
            cnt16[(int64_t)cc * gridDim.x + blockIdx.x] = (uint8_t)c;
        }
    }
    // The index array may live in pinned HOST memory (read in place by the host coder: no copy command).  A channel's PT = 16
    // pixels of this block are 16 contiguous bytes of the CHW array: one 16-byte store per channel where the rows are
    // aligned (H W a multiple of 16: 1080p, 4K), byte stores otherwise.
    if ((HW & (PT - 1)) == 0 && (reinterpret_cast<uintptr_t>(idx_chw) & 15) == 0) {
        for (int cc = threadIdx.x; cc < Cg; cc += EB)
            *reinterpret_cast<uint4*>(idx_chw + (int64_t)cc * HW + p0) = *reinterpret_cast<const uint4*>(sp + cc * PT);
        return;
    }
    for (int it = threadIdx.x; it < PT * Cg; it += EB) {
        const int cc = it / PT, pl = it - cc * PT;
        const int64_t p = p0 + pl;
        if (p < HW) idx_chw[(int64_t)cc * HW + p] = sp[cc * PT + pl];
    }
}

template <typename T>
__global__ __launch_bounds__(EB) void prior_dec_restore_kernel(int n_groups, int step, const int8_t* sym_chw, const T* mu,
                                                              int64_t ldm, int H, int W, int C, const T* hin,
                                                              int64_t ldhi, T* hout, int64_t ldho)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int8_t* sp = reinterpret_cast<int8_t*>(smem);   // [Cg][PT]
    const int Cg = C / n_groups;
    const int64_t HW = (int64_t)H * W;
    const int64_t p0 = (int64_t)blockIdx.x * PT;
    // (the symbols may live in pinned HOST memory, written there by the host coder: 16-byte loads where the rows are aligned)
    if ((HW & (PT - 1)) == 0 && (reinterpret_cast<uintptr_t>(sym_chw) & 15) == 0) {
        for (int cc = threadIdx.x; cc < Cg; cc += EB)
            *reinterpret_cast<uint4*>(sp + cc * PT) = *reinterpret_cast<const uint4*>(sym_chw + (int64_t)cc * HW + p0);
    } else {
        for (int it = threadIdx.x; it < PT * Cg; it += EB) {
            const int cc = it / PT, pl = it - cc * PT;
            const int64_t p = p0 + pl;
            sp[cc * PT + pl] = p < HW ? sym_chw[(int64_t)cc * HW + p] : (int8_t)0;
        }
    }
    __syncthreads();
    for (int it = threadIdx.x; it < PT * Cg; it += EB) {
        const int pl = it / Cg, cc = it - pl * Cg;
        const int64_t p = p0 + pl;
        if (p >= HW) continue;
        const int h = (int)p / W, w = (int)p - h * W;
        const int ga = active_group(n_groups, step, h, w);
        for (int g = 0; g < n_groups; ++g) {
            const int ch = cc + g * Cg;
            const float prev = step == 0 ? 0.f : ld(hin, p * ldhi + ch);
            if (g != ga) {
                st(hout, p * ldho + ch, prev);
                continue;
            }
            const float yh = (float)to_t<T>((float)sp[cc * PT + pl] + ld(mu, p * ldm + ch));
            st(hout, p * ldho + ch, step == 0 ? yh : prev + yh);
        }
    }
}

template <typename T>
__global__ void prior_finish_kernel(int q_mode, T* yh, int64_t ldh, const T* qs, int64_t ldq, int64_t HW, int C)
{
    const int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x;
    if (i >= HW * C) return;
    const int64_t p = (int)i / C;           // (H * W * C < 2^31, checked by the caller: 32-bit division)
    const int c = (int)i - (int)p * C;
    float q;
    if (q_mode == 0) {
        q = ld(qs, p * ldq + c);
        q = q < 0.5f ? 0.5f : q;
    } else {
        q = dcvc_sigmoidf(ld(qs, p * ldq + 1)) * 1.5f + 0.5f;
    }
    st(yh, p * ldh + c, ld(yh, p * ldh + c) * q);
}

// ------------------------------------------------------------------ operator-module (flat) kernels
template <typename T>
__global__ void op_process_with_mask_kernel(const T* y, const T* sc, const T* mu, const T* mask, float thres, T* y_res,
                                            T* y_q, T* y_hat, T* s_hat, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x;
    if (i >= n) return;
    const float mk = ld(mask, i);
    const float sh = ld(sc, i) * mk, mh = ld(mu, i) * mk;
    const float yr = (ld(y, i) - mh) * mk;
    float q = dcvc_roundf(yr);
    if (thres >= 0.f) q = q * (sh > thres ? 1.f : 0.f);
    q = clampf(q, -128.f, 127.f);
    st(y_res, i, yr);
    st(y_q, i, q);
    st(y_hat, i, q + mh);
    st(s_hat, i, sh);
}

template <typename T>
__global__ void op_combine_2x_kernel(T* out, const T* x, const T* mask, int64_t hn)
{
    const int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x;
    if (i >= hn) return;
    st(out, i, ld(x, i) * ld(mask, i) + ld(x, i + hn) * ld(mask, i + hn));
}

template <typename T>
__global__ void op_restore_kernel(T* out, const T* y, const T* mu, const T* mask, int64_t gn, int groups)
{
    const int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x;
    if (i >= gn) return;
    const float v = ld(y, i);
    for (int g = 0; g < groups; ++g) st(out, i + g * gn, (v + ld(mu, i + g * gn)) * ld(mask, i + g * gn));
}

template <typename T>
__global__ void op_build_index_kernel(int16_t* out_enc, uint8_t* out_dec, uint8_t* cond, const T* sym, const T* sc,
                                      float smin, float smax, float lmin, float lrec, float thres, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x;
    if (i >= n) return;
    const float s = clampf(ld(sc, i), smin, smax);
    const int idx = dcvc_scale_to_index(s, smin, smax, lmin, lrec);
    if (out_dec) out_dec[i] = (uint8_t)idx;
    if (out_enc) out_enc[i] = (int16_t)((int)ld(sym, i) * 256 + idx);
    if (cond) cond[i] = s > thres ? 1 : 0;
}

template <typename T>
__global__ void op_round_int8_kernel(T* z, int8_t* z8, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x;
    if (i >= n) return;
    const float v = clampf(dcvc_roundf(ld(z, i)), -128.f, 127.f);
    st(z, i, v);
    z8[i] = (int8_t)v;
}

template <typename T>
__global__ void op_clamp_recip_kernel(const T* q, T* y, float min_val, T* q_out, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x;
    if (i >= n) return;
    float qv = ld(q, i);
    qv = qv < min_val ? min_val : qv;
    st(q_out, i, qv);
    st(y, i, ld(y, i) * (1.0f / qv));
}

template <typename T>
__global__ void op_add_mul_kernel(T* x0, const T* x1, const T* q, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x;
    if (i >= n) return;
    st(x0, i, (ld(x0, i) + ld(x1, i)) * ld(q, i));
}

template <typename T>
__global__ void op_bias_quant_kernel(T* x, const T* bias, const T* q, int C, int64_t HW)
{
    const int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x;
    if (i >= HW * C) return;
    const int c = (int)(i / HW);
    st(x, i, (ld(x, i) + ld(bias, c)) * ld(q, c));
}

template <typename T>
__global__ void op_ps8_kernel(T* out, const T* x, const T* bias, int C, int H, int W, int do_clamp)
{   // x: NCHW [C][H][W], out: [C/64][8H][8W]
    const int HO = H * 8, WO = W * 8, CO = C / 64;
    const int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x;
    if (i >= (int64_t)CO * HO * WO) return;
    const int xw = (int)(i % WO), y = (int)((i / WO) % HO), c = (int)(i / ((int64_t)WO * HO));
    const int ch = c * 64 + (y & 7) * 8 + (xw & 7);
    float v = ld(x, ((int64_t)ch * H + (y >> 3)) * W + (xw >> 3)) + ld(bias, ch);
    if (do_clamp) v = clampf(v, 0.f, 1.f);
    st(out, i, v);
}

template <typename T>
__global__ void op_replicate_pad_kernel(const T* x, int C, int H, int W, int HO, int WO, T* out)
{
    const int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x;
    if (i >= (int64_t)C * HO * WO) return;
    const int xw = (int)(i % WO), y = (int)((i / WO) % HO), c = (int)(i / ((int64_t)WO * HO));
    const int sy = y < H ? y : H - 1, sx = xw < W ? xw : W - 1;
    out[i] = x[((int64_t)c * H + sy) * W + sx];
}

template <typename T>
__global__ void op_bias_wsilu_dw_kernel(const T* x, const T* w, const T* bias, int C, int H, int W, T* out)
{   // NCHW, w: [C][3][3]
    const int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x;
    if (i >= (int64_t)C * H * W) return;
    const int xw = (int)(i % W), y = (int)((i / W) % H), c = (int)(i / ((int64_t)W * H));
    const float b = ld(bias, c);
    float s = 0.f;
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = y + ky - 1;
        if (iy < 0 || iy >= H) continue;
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = xw + kx - 1;
            if (ix < 0 || ix >= W) continue;
            // activation rounded to the storage type like the reference's smem tile inputs
            const float av = (float)to_t<T>(dcvc_wsiluf(ld(x, ((int64_t)c * H + iy) * W + ix) + b));
            s = DCVC_FMAF(av, ld(w, c * 9 + ky * 3 + kx), s);
        }
    }
    st(out, i, s);
}

template <typename F>
int typed(int dtype, F&& launch)
{
    if (dtype == DCVC_F16)
        launch(half_t{});
    else if (dtype == DCVC_F32)
        launch(float{});
    else {
        dcvc::set_error("bad dtype %d", dtype);
        return dcvc::E_ARG;
    }
    DCVC_LAUNCH_CHECK();
    return 0;
}


// ------------------------------------------------------------------------------------------
// Symbol hand-off of the encoder without a copy command: the kept symbols of each part of `packed` ((sym << 8) | index,
// low byte 0xFF = skipped) are compacted IN ORDER (the order is the bit stream) and written straight into pinned host
// memory, part p at [p * n_per_part, + counts[p]).  Two launches: per-block counts, then prefix + scatter.
// (reference: the boolean-mask compaction out[skip_cond] + .cpu() of cuda_inference.py:159 / entropy_models.py:48, which
// costs a device synchronisation for the size.)
constexpr int CB = 256;                 // threads per compaction block
__global__ __launch_bounds__(CB) void compact_count_kernel(const int16_t* packed, int n_per_part, int seg, int* block_counts)
{
    const int part = blockIdx.y, b = blockIdx.x, lo = b * seg, hi = min(lo + seg, n_per_part);
    const int16_t* src = packed + (size_t)part * n_per_part;
    int c = 0;
    for (int i = lo + threadIdx.x; i < hi; i += CB) c += ((src[i] & 0xFF) != 0xFF);
    __shared__ int red[CB / 64];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_counts[part * gridDim.x + b] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(CB) void compact_scatter_kernel(const int16_t* packed, int n_per_part, int seg, const int* block_counts,
                                                             int16_t* out, int* counts)
{
    const int part = blockIdx.y, b = blockIdx.x, nb = gridDim.x, lo = b * seg, hi = min(lo + seg, n_per_part);
    const int16_t* src = packed + (size_t)part * n_per_part;
    int16_t* dst = out + (size_t)part * n_per_part;
    __shared__ int red[CB / 64];
    __shared__ int base_sh;
    // symbols kept by the blocks in front of this one
    int pre = 0;
    for (int i = threadIdx.x; i < b; i += CB) pre += block_counts[part * nb + i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) pre += __shfl_down(pre, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = pre;
    __syncthreads();
    if (threadIdx.x == 0) {
        base_sh = red[0] + red[1] + red[2] + red[3];
        if (b == nb - 1) counts[part] = base_sh + block_counts[part * nb + b];
    }
    __syncthreads();
    int base = base_sh;
    // the segment in rounds of CB consecutive symbols: ballot-based ranks keep the order
    __shared__ int wsum[CB / 64];
    for (int i0 = lo; i0 < hi; i0 += CB) {
        const int i = i0 + threadIdx.x;
        const int16_t v = i < hi ? src[i] : (int16_t)0xFF;
        const bool keep = (v & 0xFF) != 0xFF;
        const unsigned long long m = __ballot(keep);
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        const int rank = __popcll(m & ((1ull << lane) - 1ull));
        __syncthreads();     // (wsum of the previous round has been read)
        if (lane == 0) wsum[w] = __popcll(m);
        __syncthreads();
        int off = 0;
        for (int k = 0; k < w; ++k) off += wsum[k];
        if (keep) dst[base + off + rank] = v;
        base += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    }
}

// ------------------------------------------------------------------------------------------
// Decoder hand-off with the kept entries compacted on the device (round 4).  The index array of a checkerboard step is
// ~96 % sentinels at the benchmarked rates: instead of copying all of it to the host (1 MB per step at 1080p, a copy
// command) and all decoded symbols back, the kept table indexes are written IN STREAM ORDER (CHW order, the order of the
// reference's boolean-mask compaction, entropy_models.py:330-341) into pinned host memory by a kernel, the host decodes
// exactly that many symbols, and the restore step fetches them back with coalesced 16-byte reads and finds each position's
// symbol by its rank.  Unit of the scan: one (channel, 16-pixel run) = 16 consecutive CHW positions; prior_dec_index_kernel
// counts the kept entries of every run (cnt16), dec_compact_kernel turns them into offsets (off16) and writes the entries.
__global__ __launch_bounds__(CB) void dec_compact_kernel(const uint8_t* idx_chw, const uint8_t* cnt16, int n16, int nblk, int64_t HW,
                                                         uint8_t* out_host, int32_t* count_host, uint32_t* off16, int32_t* total_dev)
{
    const int b = blockIdx.x, nb = gridDim.x;
    const int per = (n16 + nb - 1) / nb;
    const int lo = min(b * per, n16), hi = min(lo + per, n16);
    __shared__ int red[CB / 64];
    __shared__ int wsum[CB / 64];
    __shared__ int base_sh;
    __shared__ __attribute__((aligned(16))) uint8_t stage[CB * PT];
    // entries kept by the runs in front of this block's (cnt16 is 4-byte aligned: whole words, then the tail)
    int pre = 0;
    const uint32_t* cw = reinterpret_cast<const uint32_t*>(cnt16);
    const int nw = lo >> 2;
    for (int i = threadIdx.x; i < nw; i += CB) {
        const uint32_t v = cw[i];
        pre += (int)((v & 0xFF) + ((v >> 8) & 0xFF) + ((v >> 16) & 0xFF) + (v >> 24));
    }
    if (threadIdx.x < (lo & 3)) pre += cnt16[(nw << 2) + threadIdx.x];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) pre += __shfl_down(pre, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = pre;
    __syncthreads();
    if (threadIdx.x == 0) base_sh = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    int base = base_sh;
    const bool wide = (HW & (PT - 1)) == 0 && (reinterpret_cast<uintptr_t>(idx_chw) & 15) == 0;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int r0 = lo; r0 < hi; r0 += CB) {
        const int r = r0 + threadIdx.x;
        const bool live = r < hi;
        const int c = live ? cnt16[r] : 0;
        // exclusive scan of c over the block: inside the wave by shuffles, across the four waves through LDS
        int inc = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(inc, o);
            if (lane >= o) inc += t;
        }
        __syncthreads();          // (stage / wsum of the previous round have been read)
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        int off = inc - c;
        for (int k = 0; k < w; ++k) off += wsum[k];
        const int round_total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        if (live) {
            off16[r] = (uint32_t)(base + off);
            if (c) {
                const int cc = r / nblk, blk = r - cc * nblk;
                const int64_t p0 = (int64_t)blk * PT;
                const uint8_t* src = idx_chw + (int64_t)cc * HW + p0;
                uint8_t v[PT];
                if (wide) {
                    *reinterpret_cast<uint4*>(v) = *reinterpret_cast<const uint4*>(src);
                } else {
                    for (int j = 0; j < PT; ++j) v[j] = p0 + j < HW ? src[j] : (uint8_t)0xFF;
                }
                int o = off;
#pragma unroll
                for (int j = 0; j < PT; ++j)
                    if (v[j] != 0xFF) stage[o++] = v[j];
            }
        }
        __syncthreads();
        // consecutive lanes write consecutive bytes of the pinned buffer: whole lines over the host link
        for (int j = threadIdx.x; j < round_total; j += CB) out_host[base + j] = stage[j];
        base += round_total;
    }
    if (b == nb - 1 && threadIdx.x == 0) {
        *count_host = base;
        *total_dev = base;
    }
}

// the decoded symbols of the step (compacted, `*total` of them, written by the host coder into pinned memory) -> device
__global__ void dec_gather_kernel(const uint4* src_host, uint4* dst, const int32_t* total_dev, int cap16)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n16 = (*total_dev + 15) >> 4;
    if (i < n16 && i < cap16) dst[i] = src_host[i];
}

// prior_dec_restore_kernel on compacted symbols: the symbol of a kept position is csym[off16[run] + rank inside the run]
template <typename T>
__global__ __launch_bounds__(EB) void prior_dec_restore_compact_kernel(int n_groups, int step, const int8_t* csym,
                                                                      const uint8_t* idx_chw, const uint32_t* off16,
                                                                      const T* mu, int64_t ldm, int H, int W, int C,
                                                                      const T* hin, int64_t ldhi, T* hout, int64_t ldho)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t* bs = reinterpret_cast<uint32_t*>(smem);            // [Cg] offset of the run's first kept entry
    uint32_t* mk = bs + C / n_groups;                            // [Cg] kept mask of the run's 16 positions
    const int Cg = C / n_groups;
    const int64_t HW = (int64_t)H * W;
    const int64_t p0 = (int64_t)blockIdx.x * PT;
    const bool wide = (HW & (PT - 1)) == 0 && (reinterpret_cast<uintptr_t>(idx_chw) & 15) == 0;
    for (int cc = threadIdx.x; cc < Cg; cc += EB) {
        const uint8_t* src = idx_chw + (int64_t)cc * HW + p0;
        uint8_t v[PT];
        if (wide) {
            *reinterpret_cast<uint4*>(v) = *reinterpret_cast<const uint4*>(src);
        } else {
            for (int j = 0; j < PT; ++j) v[j] = p0 + j < HW ? src[j] : (uint8_t)0xFF;
        }
        uint32_t m = 0;
#pragma unroll
        for (int j = 0; j < PT; ++j) m |= (uint32_t)(v[j] != 0xFF) << j;
        mk[cc] = m;
        bs[cc] = off16[(int64_t)cc * gridDim.x + blockIdx.x];
    }
    __syncthreads();
    for (int it = threadIdx.x; it < PT * Cg; it += EB) {
        const int pl = it / Cg, cc = it - pl * Cg;
        const int64_t p = p0 + pl;
        if (p >= HW) continue;
        const int h = (int)p / W, w = (int)p - h * W;
        const int ga = active_group(n_groups, step, h, w);
        const uint32_t m = mk[cc];
        const float sym = (m >> pl) & 1u ? (float)csym[bs[cc] + __popc(m & ((1u << pl) - 1u))] : 0.f;
        for (int g = 0; g < n_groups; ++g) {
            const int ch = cc + g * Cg;
            const float prev = step == 0 ? 0.f : ld(hin, p * ldhi + ch);
            if (g != ga) {
                st(hout, p * ldho + ch, prev);
                continue;
            }
            const float yh = (float)to_t<T>(sym + ld(mu, p * ldm + ch));
            st(hout, p * ldho + ch, step == 0 ? yh : prev + yh);
        }
    }
}

__global__ void copy_f32_kernel(float* dst, const float* src, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

}  // namespace

extern "C" {

int dcvc_unshuffle8(int dtype, const void* x, int C, int H, int W, void* out, int64_t ldo, void* stream)
{
    DCVC_REQUIRE(x && out && H % 8 == 0 && W % 8 == 0 && ldo >= C * 64, "dcvc_unshuffle8: bad arguments");
    return typed(dtype, [&](auto tag) {
        using T = decltype(tag);
        constexpr int V = Traits<T>::kVec;
        const size_t lds = (size_t)C * 8 * S8_PIX * 8 * sizeof(T);
        const bool tiled = W % 8 == 0 && H % 8 == 0 && (W % V) == 0 && ldo % V == 0 && lds <= 64 * 1024 &&
                           ((uintptr_t)x | (uintptr_t)out) % 16 == 0;
        if (tiled) {
            const int segs = (W / 8 + S8_PIX - 1) / S8_PIX;
            unshuffle8_tiled_kernel<T><<<(H / 8) * segs, EB, lds, (hipStream_t)stream>>>((const T*)x, C, H, W, (T*)out, ldo);
        } else {
            unshuffle8_kernel<T><<<nblocks((int64_t)C * H * W), EB, 0, (hipStream_t)stream>>>((const T*)x, C, H, W, (T*)out, ldo);
        }
    });
}

int dcvc_shuffle8_clamp(int dtype, const void* x, int64_t ld_, const float* bias, int C, int H, int W, int do_clamp,
                        void* out, void* stream)
{
    DCVC_REQUIRE(x && out && ld_ >= C * 64, "dcvc_shuffle8_clamp: bad arguments");
    return typed(dtype, [&](auto tag) {
        using T = decltype(tag);
        constexpr int V = Traits<T>::kVec;
        const size_t lds = (size_t)C * 8 * S8_PIX * 8 * sizeof(T);
        const bool tiled = ld_ % V == 0 && lds <= 64 * 1024 && ((uintptr_t)x | (uintptr_t)out) % 16 == 0;
        if (tiled) {
            const int segs = (W + S8_PIX - 1) / S8_PIX;
            shuffle8_tiled_kernel<T><<<H * segs, EB, lds, (hipStream_t)stream>>>((const T*)x, ld_, bias, C, H, W, do_clamp, (T*)out);
        } else {
            shuffle8_kernel<T><<<nblocks((int64_t)C * H * W * 64), EB, 0, (hipStream_t)stream>>>((const T*)x, ld_, bias, C, H, W, do_clamp, (T*)out);
        }
    });
}

int dcvc_replicate_pad_hwc(int dtype, const void* x, int64_t ldx, int H, int W, int C, int pad_b, int pad_r, void* out,
                           int64_t ldo, void* stream)
{
    DCVC_REQUIRE(x && out && pad_b >= 0 && pad_r >= 0 && H > 0 && W > 0, "dcvc_replicate_pad_hwc: bad arguments");
    return typed(dtype, [&](auto tag) {
        using T = decltype(tag);
        replicate_pad_hwc_kernel<T><<<nblocks((int64_t)(H + pad_b) * (W + pad_r) * C), EB, 0, (hipStream_t)stream>>>((const T*)x, ldx, H, W, C,
                     H + pad_b, W + pad_r, (T*)out, ldo);
    });
}

int dcvc_scale_channels(int dtype, const void* x, int64_t ldx, const float* q, int64_t P, int C, void* out, int64_t ldo,
                        void* stream)
{
    DCVC_REQUIRE(x && out && q, "dcvc_scale_channels: null pointer");
    return typed(dtype, [&](auto tag) {
        using T = decltype(tag);
        constexpr int V = Traits<T>::kVec;
        const bool vec = C % V == 0 && ldx % V == 0 && ldo % V == 0 && ((uintptr_t)x | (uintptr_t)out) % 16 == 0;
        if (vec)
            scale_channels_vec_kernel<T><<<nblocks(P * (C / V)), EB, 0, (hipStream_t)stream>>>((const T*)x, ldx, q, P, C, (T*)out, ldo);
        else
            scale_channels_kernel<T><<<nblocks(P * C), EB, 0, (hipStream_t)stream>>>((const T*)x, ldx, q, P, C, (T*)out, ldo);
    });
}

int dcvc_copy_channels(int dtype, const void* x, int64_t ldx, int64_t P, int C, void* out, int64_t ldo, void* stream)
{
    DCVC_REQUIRE(x && out, "dcvc_copy_channels: null pointer");
    return typed(dtype, [&](auto tag) {
        using T = decltype(tag);
        copy_channels_kernel<T><<<nblocks(P * C), EB, 0, (hipStream_t)stream>>>((const T*)x, ldx, P, C, (T*)out, ldo);
    });
}

int dcvc_crop_hwc(int dtype, const void* x, int64_t ldx, int W, int H2, int W2, int C, void* out, int64_t ldo, void* stream)
{
    DCVC_REQUIRE(x && out && W2 <= W, "dcvc_crop_hwc: bad arguments");
    return typed(dtype, [&](auto tag) {
        using T = decltype(tag);
        crop_hwc_kernel<T><<<nblocks((int64_t)H2 * W2 * C), EB, 0, (hipStream_t)stream>>>((const T*)x, ldx, W, H2, W2, C, (T*)out, ldo);
    });
}

int dcvc_nchw_to_hwc(int dtype, const void* x, int C, int64_t HW, void* out, int64_t ldo, void* stream)
{
    DCVC_REQUIRE(x && out && ldo >= C, "dcvc_nchw_to_hwc: bad arguments");
    return typed(dtype, [&](auto tag) {
        using T = decltype(tag);
        nchw_to_hwc_kernel<T><<<nblocks(HW * C), EB, 0, (hipStream_t)stream>>>((const T*)x, C, HW, (T*)out, ldo);
    });
}

int dcvc_hwc_to_nchw(int dtype, const void* x, int64_t ldx, int C, int64_t HW, void* out, void* stream)
{
    DCVC_REQUIRE(x && out && ldx >= C, "dcvc_hwc_to_nchw: bad arguments");
    return typed(dtype, [&](auto tag) {
        using T = decltype(tag);
        hwc_to_nchw_kernel<T><<<nblocks(HW * C), EB, 0, (hipStream_t)stream>>>((const T*)x, ldx, C, HW, (T*)out);
    });
}

int dcvc_round_z(int dtype, void* z, int64_t ldz, int H, int W, int C, int8_t* z_chw, void* stream)
{
    DCVC_REQUIRE(z && z_chw, "dcvc_round_z: null pointer");
    return typed(dtype, [&](auto tag) {
        using T = decltype(tag);
        round_z_kernel<T><<<nblocks((int64_t)H * W * C), EB, 0, (hipStream_t)stream>>>((T*)z, ldz, (int64_t)H * W, C, z_chw);
    });
}

int dcvc_z_from_int8(int dtype, const int8_t* z_chw, int H, int W, int C, void* out, int64_t ldo, void* stream)
{
    DCVC_REQUIRE(z_chw && out, "dcvc_z_from_int8: null pointer");
    return typed(dtype, [&](auto tag) {
        using T = decltype(tag);
        z_from_int8_kernel<T><<<nblocks((int64_t)H * W * C), EB, 0, (hipStream_t)stream>>>(z_chw, (int64_t)H * W, C, (T*)out, ldo);
    });
}

int dcvc_prior_enc_step(int dtype, int n_groups, int step, int q_mode, const void* y, int64_t ldy, const void* qsrc,
                        int64_t ldq, const void* scales, int64_t lds_, const void* means, int64_t ldm, int H, int W, int C,
                        float thres, const void* yhat_in, int64_t ldhi, void* yhat_out, int64_t ldho, int16_t* packed_chw,
                        void* stream)
{
    DCVC_REQUIRE(y && qsrc && scales && means && yhat_out && packed_chw, "dcvc_prior_enc_step: null pointer");
    DCVC_REQUIRE((int64_t)H * W < (1ll << 31), "dcvc_prior_enc_step: map too large");
    DCVC_REQUIRE((n_groups == 2 || n_groups == 4) && step >= 0 && step < n_groups && C % n_groups == 0,
                 "dcvc_prior_enc_step: bad groups/step %d/%d", n_groups, step);
    DCVC_REQUIRE(step == 0 || yhat_in, "dcvc_prior_enc_step: yhat_in required after step 0");
    PriorEncArgs a{n_groups, step, q_mode, y, ldy, qsrc, ldq, scales, lds_, means, ldm, H, W, C, thres,
                   yhat_in, ldhi, yhat_out, ldho, packed_chw};
    const int64_t HW = (int64_t)H * W;
    const int grid = (int)((HW + PT - 1) / PT);
    const size_t lds = (size_t)(C / n_groups) * PT * sizeof(int16_t);
    return typed(dtype, [&](auto tag) {
        using T = decltype(tag);
        prior_enc_kernel<T><<<grid, EB, lds, (hipStream_t)stream>>>(a);
    });
}

int dcvc_prior_dec_index(int dtype, int n_groups, int step, const void* scales, int64_t lds_, int H, int W, int C,
                         float thres, uint8_t* idx_chw, void* stream)
{
    DCVC_REQUIRE(scales && idx_chw, "dcvc_prior_dec_index: null pointer");
    DCVC_REQUIRE((int64_t)H * W < (1ll << 31), "dcvc_prior_dec_index: map too large");
    DCVC_REQUIRE((n_groups == 2 || n_groups == 4) && step >= 0 && step < n_groups && C % n_groups == 0,
                 "dcvc_prior_dec_index: bad groups/step");
    const int grid = (int)(((int64_t)H * W + PT - 1) / PT);
    const size_t lds = (size_t)(C / n_groups) * PT;
    return typed(dtype, [&](auto tag) {
        using T = decltype(tag);
        prior_dec_index_kernel<T><<<grid, EB, lds, (hipStream_t)stream>>>(n_groups,
                                    step, (const T*)scales, lds_, H, W, C, thres, idx_chw, (uint8_t*)nullptr);
    });
}

int dcvc_prior_dec_restore(int dtype, int n_groups, int step, const int8_t* sym_chw, const void* means, int64_t ldm, int H,
                           int W, int C, const void* yhat_in, int64_t ldhi, void* yhat_out, int64_t ldho, void* stream)
{
    DCVC_REQUIRE(sym_chw && means && yhat_out, "dcvc_prior_dec_restore: null pointer");
    DCVC_REQUIRE((int64_t)H * W < (1ll << 31), "dcvc_prior_dec_restore: map too large");
    DCVC_REQUIRE((n_groups == 2 || n_groups == 4) && step >= 0 && step < n_groups && C % n_groups == 0,
                 "dcvc_prior_dec_restore: bad groups/step");
    DCVC_REQUIRE(step == 0 || yhat_in, "dcvc_prior_dec_restore: yhat_in required after step 0");
    const int grid = (int)(((int64_t)H * W + PT - 1) / PT);
    const size_t lds = (size_t)(C / n_groups) * PT;
    return typed(dtype, [&](auto tag) {
        using T = decltype(tag);
        prior_dec_restore_kernel<T><<<grid, EB, lds, (hipStream_t)stream>>>(n_groups,
                                    step, sym_chw, (const T*)means, ldm, H, W, C, (const T*)yhat_in, ldhi, (T*)yhat_out,
                                    ldho);
    });
}

namespace {
// workspace of the compacted decoder hand-off (device): [total: 16 bytes][off16: n16 x 4][cnt16: n16 -> 16][csym: n -> 16]
struct DecWs {
    int64_t n16, n, off_off16, off_cnt16, off_csym, bytes;
    int nblk;
};
DecWs dec_ws(int H, int W, int C, int n_groups)
{
    DecWs w{};
    const int64_t HW = (int64_t)H * W;
    w.nblk = (int)((HW + PT - 1) / PT);
    w.n16 = (int64_t)(C / n_groups) * w.nblk;
    w.n = (int64_t)(C / n_groups) * HW;
    w.off_off16 = 16;
    w.off_cnt16 = w.off_off16 + w.n16 * 4;
    w.off_csym = w.off_cnt16 + (w.n16 + 15) / 16 * 16;
    w.bytes = w.off_csym + (w.n + 15) / 16 * 16;
    return w;
}
}  // namespace

int64_t dcvc_prior_dec_compact_ws_bytes(int H, int W, int C, int n_groups)
{
    if (H <= 0 || W <= 0 || C <= 0 || (n_groups != 2 && n_groups != 4) || C % n_groups) return 0;
    return dec_ws(H, W, C, n_groups).bytes;
}

int dcvc_prior_dec_index_compact(int dtype, int n_groups, int step, const void* scales, int64_t lds_, int H, int W, int C,
                                 float thres, uint8_t* idx_chw, void* workspace, uint8_t* idx_host, int32_t* count_host,
                                 void* stream)
{
    DCVC_REQUIRE(scales && idx_chw && workspace && idx_host && count_host, "dcvc_prior_dec_index_compact: null pointer");
    DCVC_REQUIRE((int64_t)H * W < (1ll << 31), "dcvc_prior_dec_index_compact: map too large");
    DCVC_REQUIRE((n_groups == 2 || n_groups == 4) && step >= 0 && step < n_groups && C % n_groups == 0,
                 "dcvc_prior_dec_index_compact: bad groups/step");
    DCVC_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "dcvc_prior_dec_index_compact: workspace must be 16-byte aligned");
    const DecWs w = dec_ws(H, W, C, n_groups);
    DCVC_REQUIRE(w.n < (1ll << 31), "dcvc_prior_dec_index_compact: too many symbols");
    char* ws = (char*)workspace;
    uint8_t* cnt16 = (uint8_t*)(ws + w.off_cnt16);
    uint8_t* out_dev = nullptr;
    int32_t* count_dev = nullptr;
    DCVC_HIP(hipHostGetDevicePointer((void**)&out_dev, idx_host, 0));
    DCVC_HIP(hipHostGetDevicePointer((void**)&count_dev, count_host, 0));
    const size_t lds = (size_t)(C / n_groups) * PT;
    int rc = typed(dtype, [&](auto tag) {
        using T = decltype(tag);
        prior_dec_index_kernel<T><<<w.nblk, EB, lds, (hipStream_t)stream>>>(n_groups,
                                    step, (const T*)scales, lds_, H, W, C, thres, idx_chw, cnt16);
    });
    if (rc) return rc;
    hipLaunchKernelGGL(dec_compact_kernel, dim3(DCVC_COMPACT_BLOCKS), dim3(CB), 0, (hipStream_t)stream, idx_chw, cnt16, (int)w.n16,
                       w.nblk, (int64_t)H * W, out_dev, count_dev, (uint32_t*)(ws + w.off_off16), (int32_t*)ws);
    DCVC_LAUNCH_CHECK();
    return 0;
}

int dcvc_prior_dec_restore_compact(int dtype, int n_groups, int step, const int8_t* sym_host, const uint8_t* idx_chw,
                                   void* workspace, const void* means, int64_t ldm, int H, int W, int C,
                                   const void* yhat_in, int64_t ldhi, void* yhat_out, int64_t ldho, void* stream)
{
    DCVC_REQUIRE(sym_host && idx_chw && workspace && means && yhat_out, "dcvc_prior_dec_restore_compact: null pointer");
    DCVC_REQUIRE((int64_t)H * W < (1ll << 31), "dcvc_prior_dec_restore_compact: map too large");
    DCVC_REQUIRE((n_groups == 2 || n_groups == 4) && step >= 0 && step < n_groups && C % n_groups == 0,
                 "dcvc_prior_dec_restore_compact: bad groups/step");
    DCVC_REQUIRE(step == 0 || yhat_in, "dcvc_prior_dec_restore_compact: yhat_in required after step 0");
    const DecWs w = dec_ws(H, W, C, n_groups);
    char* ws = (char*)workspace;
    const int8_t* src_dev = nullptr;
    DCVC_HIP(hipHostGetDevicePointer((void**)&src_dev, (void*)sym_host, 0));
    DCVC_REQUIRE((reinterpret_cast<uintptr_t>(src_dev) & 15) == 0, "dcvc_prior_dec_restore_compact: symbol buffer must be 16-byte aligned");
    const int cap16 = (int)((w.n + 15) / 16);
    hipLaunchKernelGGL(dec_gather_kernel, dim3((cap16 + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const uint4*)src_dev,
                       (uint4*)(ws + w.off_csym), (const int32_t*)ws, cap16);
    const size_t lds = (size_t)(C / n_groups) * 8;
    return typed(dtype, [&](auto tag) {
        using T = decltype(tag);
        prior_dec_restore_compact_kernel<T><<<w.nblk, EB, lds, (hipStream_t)stream>>>(n_groups, step,
                                    (const int8_t*)(ws + w.off_csym), idx_chw, (const uint32_t*)(ws + w.off_off16),
                                    (const T*)means, ldm, H, W, C, (const T*)yhat_in, ldhi, (T*)yhat_out, ldho);
    });
}

int dcvc_prior_finish(int dtype, int q_mode, void* yhat, int64_t ldh, const void* qsrc, int64_t ldq, int H, int W, int C,
                      void* stream)
{
    DCVC_REQUIRE(yhat && qsrc, "dcvc_prior_finish: null pointer");
    DCVC_REQUIRE((int64_t)H * W * C < (1ll << 31), "dcvc_prior_finish: latent too large");
    return typed(dtype, [&](auto tag) {
        using T = decltype(tag);
        prior_finish_kernel<T><<<nblocks((int64_t)H * W * C), EB, 0, (hipStream_t)stream>>>(q_mode, (T*)yhat, ldh, (const T*)qsrc, ldq,
                     (int64_t)H * W, C);
    });
}


int dcvc_yuv420_to_frame(int dtype, const uint8_t* y, const uint8_t* u, const uint8_t* v, int H, int W, int pad_b,
                         int pad_r, void* out_nchw, void* stream)
{
    DCVC_REQUIRE(y && u && v && out_nchw && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && pad_b >= 0 && pad_r >= 0,
                 "dcvc_yuv420_to_frame: bad arguments (%dx%d)", H, W);
    return typed(dtype, [&](auto tag) {
        using T = decltype(tag);
        yuv420_to_frame_kernel<T><<<nblocks((int64_t)3 * (H + pad_b) * (W + pad_r)), EB, 0, (hipStream_t)stream>>>(
            y, u, v, H, W, H + pad_b, W + pad_r, (T*)out_nchw);
    });
}

int dcvc_frame_to_yuv420(int dtype, const void* x_nchw, int Hp, int Wp, int H, int W, int round_uv, uint8_t* y,
                         uint8_t* u, uint8_t* v, void* stream)
{
    DCVC_REQUIRE(x_nchw && y && u && v && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && Hp >= H && Wp >= W,
                 "dcvc_frame_to_yuv420: bad arguments");
    return typed(dtype, [&](auto tag) {
        using T = decltype(tag);
        frame_to_yuv420_kernel<T><<<nblocks((int64_t)H * W * 3 / 2), EB, 0, (hipStream_t)stream>>>(
            (const T*)x_nchw, Hp, Wp, H, W, round_uv, y, u, v);
    });
}

int dcvc_rgb_to_frame(int dtype, const uint8_t* rgb, int H, int W, int pad_b, int pad_r, void* out_nchw, void* stream)
{
    DCVC_REQUIRE(rgb && out_nchw && H > 0 && W > 0 && pad_b >= 0 && pad_r >= 0, "dcvc_rgb_to_frame: bad arguments (%dx%d)", H, W);
    return typed(dtype, [&](auto tag) {
        using T = decltype(tag);
        rgb_to_frame_kernel<T><<<nblocks((int64_t)(H + pad_b) * (W + pad_r)), EB, 0, (hipStream_t)stream>>>(
            rgb, H, W, H + pad_b, W + pad_r, (T*)out_nchw);
    });
}

int dcvc_frame_to_rgb(int dtype, const void* x_nchw, int Hp, int Wp, int H, int W, void* out_chw, void* stream)
{
    DCVC_REQUIRE(x_nchw && out_chw && H > 0 && W > 0 && Hp >= H && Wp >= W, "dcvc_frame_to_rgb: bad arguments");
    return typed(dtype, [&](auto tag) {
        using T = decltype(tag);
        frame_to_rgb_kernel<T><<<nblocks((int64_t)H * W), EB, 0, (hipStream_t)stream>>>((const T*)x_nchw, Hp, Wp, H, W, (T*)out_chw);
    });
}

// ---------------------------------------------------------------- operator-module seam
int dcvc_op_process_with_mask(int dtype, const void* y, const void* scales, const void* means, const void* mask,
                              float thres, void* y_res, void* y_q, void* y_hat, void* s_hat, int64_t n, void* stream)
{
    DCVC_REQUIRE(y && scales && means && mask && y_res && y_q && y_hat && s_hat, "dcvc_op_process_with_mask: null pointer");
    return typed(dtype, [&](auto tag) {
        using T = decltype(tag);
        op_process_with_mask_kernel<T><<<nblocks(n), EB, 0, (hipStream_t)stream>>>((const T*)y, (const T*)scales, (const T*)means,
                     (const T*)mask, thres, (T*)y_res, (T*)y_q, (T*)y_hat, (T*)s_hat, n);
    });
}

int dcvc_op_combine_for_reading_2x(int dtype, void* out, const void* x, const void* mask, int64_t half_n, void* stream)
{
    DCVC_REQUIRE(out && x && mask, "dcvc_op_combine_for_reading_2x: null pointer");
    return typed(dtype, [&](auto tag) {
        using T = decltype(tag);
        op_combine_2x_kernel<T><<<nblocks(half_n), EB, 0, (hipStream_t)stream>>>((T*)out, (const T*)x, (const T*)mask, half_n);
    });
}

int dcvc_op_restore_y_2x(int dtype, void* out, const void* y, const void* means, const void* mask, int64_t half_n,
                         void* stream)
{
    DCVC_REQUIRE(out && y && means && mask, "dcvc_op_restore_y_2x: null pointer");
    return typed(dtype, [&](auto tag) {
        using T = decltype(tag);
        op_restore_kernel<T><<<nblocks(half_n), EB, 0, (hipStream_t)stream>>>((T*)out, (const T*)y, (const T*)means, (const T*)mask, half_n, 2);
    });
}

int dcvc_op_restore_y_4x(int dtype, void* out, const void* y, const void* means, const void* mask, int64_t quarter_n,
                         void* stream)
{
    DCVC_REQUIRE(out && y && means && mask, "dcvc_op_restore_y_4x: null pointer");
    return typed(dtype, [&](auto tag) {
        using T = decltype(tag);
        op_restore_kernel<T><<<nblocks(quarter_n), EB, 0, (hipStream_t)stream>>>((T*)out, (const T*)y, (const T*)means, (const T*)mask, quarter_n, 4);
    });
}

int dcvc_op_build_index_dec(int dtype, uint8_t* out, uint8_t* cond_out, const void* scales, float scale_min,
                            float scale_max, float log_scale_min, float log_step_recip, float skip_thres, int64_t n,
                            void* stream)
{
    DCVC_REQUIRE(out && scales, "dcvc_op_build_index_dec: null pointer");
    return typed(dtype, [&](auto tag) {
        using T = decltype(tag);
        op_build_index_kernel<T><<<nblocks(n), EB, 0, (hipStream_t)stream>>>((int16_t*)nullptr, out, cond_out, (const T*)nullptr,
                     (const T*)scales, scale_min, scale_max, log_scale_min, log_step_recip, skip_thres, n);
    });
}

int dcvc_op_build_index_enc(int dtype, int16_t* out, uint8_t* cond_out, const void* symbols, const void* scales,
                            float scale_min, float scale_max, float log_scale_min, float log_step_recip,
                            float skip_thres, int64_t n, void* stream)
{
    DCVC_REQUIRE(out && symbols && scales, "dcvc_op_build_index_enc: null pointer");
    return typed(dtype, [&](auto tag) {
        using T = decltype(tag);
        op_build_index_kernel<T><<<nblocks(n), EB, 0, (hipStream_t)stream>>>(out, (uint8_t*)nullptr, cond_out, (const T*)symbols,
                     (const T*)scales, scale_min, scale_max, log_scale_min, log_step_recip, skip_thres, n);
    });
}

int dcvc_op_round_and_to_int8(int dtype, void* z, int8_t* z_int8, int64_t n, void* stream)
{
    DCVC_REQUIRE(z && z_int8, "dcvc_op_round_and_to_int8: null pointer");
    return typed(dtype, [&](auto tag) {
        using T = decltype(tag);
        op_round_int8_kernel<T><<<nblocks(n), EB, 0, (hipStream_t)stream>>>((T*)z, z_int8, n);
    });
}

int dcvc_op_clamp_reciprocal_with_quant(int dtype, const void* q_dec, void* y, float min_val, void* q_out, int64_t n,
                                        void* stream)
{
    DCVC_REQUIRE(q_dec && y && q_out, "dcvc_op_clamp_reciprocal_with_quant: null pointer");
    return typed(dtype, [&](auto tag) {
        using T = decltype(tag);
        op_clamp_recip_kernel<T><<<nblocks(n), EB, 0, (hipStream_t)stream>>>((const T*)q_dec, (T*)y, min_val, (T*)q_out, n);
    });
}

int dcvc_op_add_and_multiply(int dtype, void* x0, const void* x1, const void* q, int64_t n, void* stream)
{
    DCVC_REQUIRE(x0 && x1 && q, "dcvc_op_add_and_multiply: null pointer");
    return typed(dtype, [&](auto tag) {
        using T = decltype(tag);
        op_add_mul_kernel<T><<<nblocks(n), EB, 0, (hipStream_t)stream>>>((T*)x0, (const T*)x1, (const T*)q, n);
    });
}

int dcvc_op_bias_quant(int dtype, void* x, const void* bias, const void* quant, int C, int64_t HW, void* stream)
{
    DCVC_REQUIRE(x && bias && quant, "dcvc_op_bias_quant: null pointer");
    return typed(dtype, [&](auto tag) {
        using T = decltype(tag);
        op_bias_quant_kernel<T><<<nblocks(HW * C), EB, 0, (hipStream_t)stream>>>((T*)x, (const T*)bias, (const T*)quant, C, HW);
    });
}

int dcvc_op_bias_pixel_shuffle_8(int dtype, void* out, const void* x, const void* bias, int C, int H, int W, int do_clamp,
                                 void* stream)
{
    DCVC_REQUIRE(out && x && bias && C % 64 == 0, "dcvc_op_bias_pixel_shuffle_8: bad arguments");
    return typed(dtype, [&](auto tag) {
        using T = decltype(tag);
        op_ps8_kernel<T><<<nblocks((int64_t)C * H * W), EB, 0, (hipStream_t)stream>>>((T*)out, (const T*)x, (const T*)bias, C, H, W, do_clamp);
    });
}

int dcvc_op_replicate_pad(int dtype, const void* x, int C, int H, int W, int pad_b, int pad_r, void* out, void* stream)
{
    DCVC_REQUIRE(x && out && pad_b >= 0 && pad_r >= 0, "dcvc_op_replicate_pad: bad arguments");
    return typed(dtype, [&](auto tag) {
        using T = decltype(tag);
        op_replicate_pad_kernel<T><<<nblocks((int64_t)C * (H + pad_b) * (W + pad_r)), EB, 0, (hipStream_t)stream>>>((const T*)x, C, H, W,
                     H + pad_b, W + pad_r, (T*)out);
    });
}

int dcvc_op_bias_wsilu_depthwise_conv2d(int dtype, const void* x, const void* weight, const void* bias, int C, int H,
                                        int W, void* out, void* stream)
{
    DCVC_REQUIRE(x && weight && bias && out, "dcvc_op_bias_wsilu_depthwise_conv2d: null pointer");
    return typed(dtype, [&](auto tag) {
        using T = decltype(tag);
        op_bias_wsilu_dw_kernel<T><<<nblocks((int64_t)C * H * W), EB, 0, (hipStream_t)stream>>>((const T*)x, (const T*)weight, (const T*)bias, C,
                     H, W, (T*)out);
    });
}

int dcvc_compact_symbols(const int16_t* packed, int n_per_part, int n_parts, int16_t* out_host, int32_t* counts_host,
                         int32_t* workspace, void* stream)
{
    DCVC_REQUIRE(packed && out_host && counts_host && workspace, "dcvc_compact_symbols: null pointer");
    DCVC_REQUIRE(n_per_part > 0 && n_parts > 0 && n_parts <= 8, "dcvc_compact_symbols: bad sizes %d x %d", n_parts, n_per_part);
    const int nb = DCVC_COMPACT_BLOCKS, seg = (n_per_part + nb - 1) / nb;
    hipStream_t st = (hipStream_t)stream;
    int16_t* out_dev = nullptr;
    int32_t* counts_dev = nullptr;
    DCVC_HIP(hipHostGetDevicePointer((void**)&out_dev, out_host, 0));
    DCVC_HIP(hipHostGetDevicePointer((void**)&counts_dev, counts_host, 0));
    hipLaunchKernelGGL(compact_count_kernel, dim3(nb, n_parts), dim3(CB), 0, st, packed, n_per_part, seg, workspace);
    hipLaunchKernelGGL(compact_scatter_kernel, dim3(nb, n_parts), dim3(CB), 0, st, packed, n_per_part, seg, workspace, out_dev,
                       counts_dev);
    DCVC_LAUNCH_CHECK();
    return 0;
}

int dcvc_copy_f32(float* dst, const float* src, int n, void* stream)
{
    DCVC_REQUIRE(dst && src && n >= 0, "dcvc_copy_f32: bad arguments");
    if (n == 0) return 0;
    hipLaunchKernelGGL(copy_f32_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, dst, src, n);
    DCVC_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
