/* dcvc_math.h - scalar fp32 math shared by the HIP kernels (device) and by host C code.
 *
 * Every function here is built ONLY from IEEE-754 correctly rounded primitives
 * (add, mul, div, fma, round-to-nearest-even) and integer bit manipulation, so that a
 * gfx950 kernel compiled with `-ffp-contract=off` and x86 C code compiled with
 * `-ffp-contract=off -mfma` produce bit-identical results.  This is what makes the
 * fp32 "exact" mode of the HIP path reproducible on a CPU ("Numerics").
 *
 * Reference functions these replace in the fp32 path:
 *   WSiLU            src/layers/layers.py:11-16      x * sigmoid(4x)
 *   wsilu (CUDA)     src/layers/extensions/inference/common.h:282-290
 *   scale_to_index   src/layers/extensions/inference/kernel.cu:280-287 / cuda_inference.py:138-140
 *   sigmoid (q_enc/q_dec of the intra model)  src/models/common_model.py:70-71
 */
#ifndef DCVC_MATH_H
#define DCVC_MATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define DCVC_HD __host__ __device__ __forceinline__
#else
#define DCVC_HD static inline
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define DCVC_FMAF(a, b, c) __builtin_fmaf((a), (b), (c))
#define DCVC_RINTF(a) __builtin_rintf((a))
#else
#define DCVC_FMAF(a, b, c) __builtin_fmaf((a), (b), (c))
#define DCVC_RINTF(a) __builtin_rintf((a))
#endif

DCVC_HD float dcvc_bits_to_float(uint32_t u)
{
    union { uint32_t u; float f; } c;
    c.u = u;
    return c.f;
}

DCVC_HD uint32_t dcvc_float_to_bits(float f)
{
    union { uint32_t u; float f; } c;
    c.f = f;
    return c.u;
}

/* exp(x), |rel err| ~ 1e-7, x clamped to [-87, 88] (all results are normal floats). */
DCVC_HD float dcvc_expf(float x)
{
    x = x > 88.0f ? 88.0f : x;
    x = x < -87.0f ? -87.0f : x;
    const float n = DCVC_RINTF(x * 1.44269504088896341f);
    float r = DCVC_FMAF(n, -0.693359375f, x);           /* ln2 high part (exact in 10 bits) */
    r = DCVC_FMAF(n, 2.12194440e-4f, r);                /* -(ln2 - high)                     */
    float p = 1.9875691500e-4f;
    p = DCVC_FMAF(p, r, 1.3981999507e-3f);
    p = DCVC_FMAF(p, r, 8.3334519073e-3f);
    p = DCVC_FMAF(p, r, 4.1665795894e-2f);
    p = DCVC_FMAF(p, r, 1.6666665459e-1f);
    p = DCVC_FMAF(p, r, 5.0000001201e-1f);
    const float rr = r * r;
    p = DCVC_FMAF(p, rr, r);
    p = p + 1.0f;
    const int32_t e = (int32_t)n;                        /* in [-126, 127] */
    return p * dcvc_bits_to_float((uint32_t)(e + 127) << 23);
}

/* natural log for normal positive x. */
DCVC_HD float dcvc_logf(float x)
{
    uint32_t u = dcvc_float_to_bits(x);
    int32_t e = (int32_t)(u >> 23) - 126;                /* x = m * 2^e, m in [0.5, 1) */
    float m = dcvc_bits_to_float((u & 0x007fffffu) | 0x3f000000u);
    if (m < 0.70710678118654752f) {
        e -= 1;
        m = m + m;
    }
    const float f = m - 1.0f;                            /* f in [-0.2929, 0.4142) */
    const float z = f * f;
    float p = 7.0376836292e-2f;
    p = DCVC_FMAF(p, f, -1.1514610310e-1f);
    p = DCVC_FMAF(p, f, 1.1676998740e-1f);
    p = DCVC_FMAF(p, f, -1.2420140846e-1f);
    p = DCVC_FMAF(p, f, 1.4249322787e-1f);
    p = DCVC_FMAF(p, f, -1.6668057665e-1f);
    p = DCVC_FMAF(p, f, 2.0000714765e-1f);
    p = DCVC_FMAF(p, f, -2.4999993993e-1f);
    p = DCVC_FMAF(p, f, 3.3333331174e-1f);
    float y = (f * z) * p;
    const float fe = (float)e;
    y = DCVC_FMAF(fe, -2.12194440e-4f, y);
    y = DCVC_FMAF(z, -0.5f, y);
    float r = f + y;
    r = DCVC_FMAF(fe, 0.693359375f, r);
    return r;
}

DCVC_HD float dcvc_sigmoidf(float x)
{
    return 1.0f / (1.0f + dcvc_expf(-x));
}

/* WSiLU(x) = sigmoid(4x) * x                      (layers.py:16) */
DCVC_HD float dcvc_wsiluf(float x)
{
    return dcvc_sigmoidf(4.0f * x) * x;
}

/* round half to even (torch.round / __half2int_rn, common.h:83-86) */
DCVC_HD float dcvc_roundf(float x)
{
    return DCVC_RINTF(x);
}

/* Gaussian scale -> CDF-table index, truncation toward zero into uint8
 * (cuda_inference.py:138-140, kernel.cu:280-287; constants entropy_models.py:230-238). */
DCVC_HD uint8_t dcvc_scale_to_index(float scale, float scale_min, float scale_max,
                                    float log_scale_min, float log_step_recip)
{
    scale = scale < scale_min ? scale_min : scale;
    scale = scale > scale_max ? scale_max : scale;
    const float v = (dcvc_logf(scale) - log_scale_min) * log_step_recip;
    int32_t i = (int32_t)v;                              /* trunc */
    i = i < 0 ? 0 : i;
    i = i > 255 ? 255 : i;
    return (uint8_t)i;
}

#endif /* DCVC_MATH_H */
