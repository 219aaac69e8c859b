"""Deterministic synthetic checkpoints (there is no network, so the reference's pretrained
``cvpr2025_{image,video}.pth.tar`` cannot be fetched - SURVEY.md section 8c).

``make_state_dict('dmc'|'dmci', seed)`` returns ``{name: float32 ndarray}`` with exactly the
reference's state_dict keys and shapes (opendcvc_amd/arch.py).  The generator is numpy PCG64, so
the GPU box regenerates identical weights and no weights need to be shipped.  Gains are chosen so
that activations stay O(1) through the 10-25 residual blocks (usable in fp16 as well).
"""
import numpy as np

from . import arch


# Output heads of the prior networks: per output-channel chunk (count, bias mean, bias std), so
# that the predicted Gaussian scales / means / quantisation steps land in the range a trained
# model produces (|means| < 1, q_dec around 1, and - like a trained model at a mid rate point -
# most predicted scales below the force-zero threshold: about a quarter of the latent symbols
# survive `scale > 0.12` and are entropy coded, the rest are skipped).  With plain random
# heads almost every symbol would be escape-coded and the reference coder's one-byte-per-symbol
# scratch buffer (rans.cpp:221) overflows.
_HEAD_BIAS = {
    "dmc": {
        "y_prior_fusion.conv.3": [(128, 1.0, 0.2), (128, -0.05, 0.2), (128, 0.0, 0.1)],
        "y_spatial_prior.conv.2": [(128, -0.05, 0.2), (128, 0.0, 0.1)],
    },
    "dmci": {
        "y_prior_fusion.3": [(2, 0.0, 0.3), (256, -0.05, 0.2), (256, 0.0, 0.1)],
        "y_spatial_prior.3": [(256, -0.05, 0.2), (256, 0.0, 0.1)],
    },
}
_GAIN = {"encoder.down": 0.2, "enc.enc_2.6": 0.2, "decoder.conv2": 0.35, "feature_adaptor_p": 0.7}


def make_state_dict(model, seed=1234, q_ramp=False):
    """q_ramp: the per-qp quantisation tables move with qp over a 16:1 range - a per-channel factor in [0.7, 1.4] times
    2^(+-(qp - 31.5) / 15.75), rising for the encoder-side tables, falling for the decoder-side ones (names with "dec" /
    "recon"), like a trained model's - instead of independent draws from [0.5, 1.5] per (qp, channel): rate points that
    really move with qp (tests/golden/sweep_ramp.json).  The other tensors are drawn exactly as without it."""
    spec = arch.spec_for(model)
    rng = np.random.Generator(np.random.PCG64(seed))
    heads = _HEAD_BIAS[model]
    sd = {}
    for name, shape, kind in spec.items:
        layer = name.rsplit(".", 1)[0]
        if layer in heads and kind in ("w", "b"):
            if kind == "w":
                v = rng.standard_normal(shape) * (0.15 / np.sqrt(shape[1]))
            else:
                v = np.concatenate([rng.standard_normal(n) * sd_ + mu for n, mu, sd_ in heads[layer]])
        elif kind in ("w", "w_res"):
            fan_in = shape[1] * shape[2] * shape[3]
            gain = 0.25 if kind == "w_res" else _GAIN.get(layer, 1.0)
            v = rng.standard_normal(shape) * (gain / np.sqrt(fan_in))
        elif kind == "dw":
            v = rng.standard_normal(shape) * (1.0 / 3.0)
        elif kind == "b":
            v = rng.standard_normal(shape) * 0.05
        elif kind == "q":
            v = rng.uniform(0.5, 1.5, shape)          # (drawn either way: the stream of random numbers stays the same)
            if q_ramp:
                per_channel = 0.7 + 0.7 * (v[:1] - 0.5)
                qp = np.arange(shape[0], dtype=np.float64).reshape((-1,) + (1,) * (len(shape) - 1))
                sign = -1.0 if ("dec" in name or "recon" in name) else 1.0
                v = per_channel * 2.0 ** (sign * (qp - 31.5) / 15.75)
        elif kind == "bitparm":
            v = rng.standard_normal(shape) * 0.5
        else:
            raise ValueError(kind)
        sd[name] = np.ascontiguousarray(v, dtype=np.float32)
    return sd


def synthetic_frame_yuv444(height, width, frame_idx=0, seed=0):
    """Smooth moving texture + grain, float32 [1,3,H,W] in [0,1] (YCbCr 4:4:4 after nearest chroma
    upsampling) - the synthetic input recipe of SURVEY.md section 8d."""
    rng = np.random.Generator(np.random.PCG64(seed))

    def plane(h, w, sigma, shift):
        gh, gw = h // 16 + 3, w // 16 + 3
        g = rng.standard_normal((gh, gw)) * sigma + 128.0
        ys = (np.arange(h) + shift[0]) / 16.0
        xs = (np.arange(w) + shift[1]) / 16.0
        y0 = np.clip(np.floor(ys).astype(int), 0, gh - 2)
        x0 = np.clip(np.floor(xs).astype(int), 0, gw - 2)
        fy = (ys - y0)[:, None]
        fx = (xs - x0)[None, :]
        a = g[y0][:, x0]
        b = g[y0][:, x0 + 1]
        c = g[y0 + 1][:, x0]
        d = g[y0 + 1][:, x0 + 1]
        return (a * (1 - fy) * (1 - fx) + b * (1 - fy) * fx + c * fy * (1 - fx) + d * fy * fx)

    sy = ((2 * frame_idx) % 16, (1 * frame_idx) % 16)
    Y = plane(height, width, 40.0, sy)
    U = plane(height // 2, width // 2, 20.0, (sy[0] / 2, sy[1] / 2))
    V = plane(height // 2, width // 2, 20.0, (sy[0] / 2, sy[1] / 2))
    grain = np.random.Generator(np.random.PCG64(seed * 7919 + frame_idx + 1))
    Y = np.clip(np.round(Y + grain.standard_normal(Y.shape) * 2.0), 0, 255)
    U = np.clip(np.round(U + grain.standard_normal(U.shape) * 2.0), 0, 255)
    V = np.clip(np.round(V + grain.standard_normal(V.shape) * 2.0), 0, 255)
    U = np.repeat(np.repeat(U, 2, axis=0), 2, axis=1)[:height, :width]
    V = np.repeat(np.repeat(V, 2, axis=0), 2, axis=1)[:height, :width]
    x = np.stack([Y, U, V], axis=0)[None].astype(np.float32) / np.float32(255.0)
    return np.ascontiguousarray(x)


def synthetic_frame_yuv420(height, width, frame_idx=0, seed=0):
    """The same synthetic frame as planar 8-bit YUV 4:2:0 (uint8 Y [H,W], U, V [H/2,W/2]) - the form the reference
    reads from disk (src/utils/video_reader.py:50-90).  synthetic_frame_yuv444 is exactly this frame after the
    reference's nearest chroma upsampling and /255."""
    x = synthetic_frame_yuv444(height, width, frame_idx, seed)[0]
    q = np.clip(np.rint(x * np.float32(255.0)), 0, 255).astype(np.uint8)
    return q[0], np.ascontiguousarray(q[1, ::2, ::2]), np.ascontiguousarray(q[2, ::2, ::2])
