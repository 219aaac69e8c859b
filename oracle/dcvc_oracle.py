"""dcvc_oracle.py - TEST INFRASTRUCTURE ONLY.

CPU restatement (numpy + the plain-C kernels of nn_oracle.c / rans_oracle.c) of the DCVC-RT
per-frame codec path.  It is the parity checker for the HIP path: only tests/, smoke() and the
cpu_baseline leg of bench.py may import it; the product (opendcvc_amd/) never does.

Pinned against the reference itself: tests/golden/make_golden.py imports the reference Python
models in the build container (torch CPU fallback ops + the reference's compiled rANS), runs
them on seeded weights/inputs and commits the outputs as fixtures; tests/test_oracle_*.py check
this file against those fixtures (rANS / CDF tables / container bit-exact, tensors to fp32
rounding).

Reference functions restated here (paths relative to /root/reference):
  DMC.compress / decompress / DPB            src/models/video_model.py:253-379
  FeatureExtractor/Encoder/Decoder/...       src/models/video_model.py:25-216 (forward_torch paths)
  DMCI.compress / decompress, Intra nets     src/models/image_model.py:17-209
  CompressionModel (masks, 2x / 4x prior)    src/models/common_model.py:36-296
  GaussianEncoder / BitEstimator / EntropyCoder  src/models/entropy_models.py:11-341
  fallback ops                               src/layers/cuda_inference.py:26-203 (else-branches)
Tensors are float32 [H, W, C] internally; the public API takes / returns NCHW like the reference.
"""
import ctypes
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def lib():
    """Loads (building if necessary) oracle/_build/liboracle.so."""
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "_build", "liboracle.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", _HERE, "_build/liboracle.so"])
        L = ctypes.CDLL(so)
        L.orc_coder_new.restype = ctypes.c_void_p
        L.orc_enc_stream.restype = ctypes.c_void_p
        _LIB = L
    return _LIB


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# --------------------------------------------------------------------------- operators (HWC)

def conv1x1(x, w, b):
    H, W, K = x.shape
    N = w.shape[0]
    x = _f32(x)
    w2 = _f32(w.reshape(N, K))
    out = np.empty((H, W, N), np.float32)
    lib().orc_conv1x1(_ptr(x), ctypes.c_int64(K), _ptr(w2), _ptr(_f32(b)) if b is not None else None,
                      _ptr(out), ctypes.c_int64(N), ctypes.c_int64(H * W), K, N)
    return out


def conv2d(x, w, b, stride, pad):
    H, W, C = x.shape
    N, _, KH, KW = w.shape
    Ho = (H + 2 * pad - KH) // stride + 1
    Wo = (W + 2 * pad - KW) // stride + 1
    out = np.empty((Ho, Wo, N), np.float32)
    lib().orc_conv2d(_ptr(_f32(x)), H, W, C, _ptr(_f32(w)), _ptr(_f32(b)), _ptr(out), N, KH, KW,
                     stride, pad)
    return out


def wsilu(x):
    x = _f32(x)
    out = np.empty_like(x)
    lib().orc_wsilu(_ptr(x), _ptr(out), ctypes.c_int64(x.size))
    return out


def sigmoid(x):
    x = _f32(x)
    out = np.empty_like(x)
    lib().orc_sigmoid(_ptr(x), _ptr(out), ctypes.c_int64(x.size))
    return out


def scale_to_index(s, smin, smax, log_smin, log_step_recip):
    s = _f32(s)
    out = np.empty(s.shape, np.uint8)
    lib().orc_scale_to_index(_ptr(s), _ptr(out), ctypes.c_int64(s.size), ctypes.c_float(smin),
                             ctypes.c_float(smax), ctypes.c_float(log_smin),
                             ctypes.c_float(log_step_recip))
    return out


def pixel_unshuffle(x, r):
    """[H, W, C] -> [H/r, W/r, C*r*r] with channel index c*r*r + dy*r + dx (torch semantics)."""
    H, W, C = x.shape
    x = x.reshape(H // r, r, W // r, r, C)            # h dy w dx c
    x = x.transpose(0, 2, 4, 1, 3)                     # h w c dy dx
    return np.ascontiguousarray(x.reshape(H // r, W // r, C * r * r))


def pixel_shuffle(x, r):
    H, W, Crr = x.shape
    C = Crr // (r * r)
    x = x.reshape(H, W, C, r, r)                       # h w c dy dx
    x = x.transpose(0, 3, 1, 4, 2)                     # h dy w dx c
    return np.ascontiguousarray(x.reshape(H * r, W * r, C))


def replicate_pad(x, pad_b, pad_r):
    if pad_b == 0 and pad_r == 0:
        return x
    return np.ascontiguousarray(np.pad(x, ((0, pad_b), (0, pad_r), (0, 0)), mode="edge"))


def nchw_to_hwc(x):
    assert x.shape[0] == 1
    return np.ascontiguousarray(np.asarray(x, np.float32)[0].transpose(1, 2, 0))


def hwc_to_nchw(x):
    return np.ascontiguousarray(x.transpose(2, 0, 1)[None])


class Net:
    """Weight dictionary + the layer primitives built on it."""

    def __init__(self, sd):
        self.sd = {k: _f32(np.asarray(v)) for k, v in sd.items()}

    def has(self, k):
        return k in self.sd

    def conv(self, x, prefix, stride=1, pad=0):
        w = self.sd[prefix + ".weight"]
        b = self.sd[prefix + ".bias"]
        if w.shape[2] == 1 and stride == 1:
            return conv1x1(x, w, b)
        return conv2d(x, w, b, stride, pad)

    def dcb(self, x, prefix, shortcut=False, q=None):
        """DepthConvBlock.forward_torch (layers.py:92-106); q is a per-channel vector."""
        sd = self.sd
        H, W, Cin = x.shape
        C = sd[prefix + ".dc.0.weight"].shape[0]
        has_ad = (prefix + ".adaptor.weight") in sd
        x = _f32(x)
        out = np.empty((H, W, C), np.float32)
        g = lambda n: _ptr(sd[prefix + n])
        qv = _f32(q).reshape(-1) if q is not None else None
        lib().orc_dcb(_ptr(x), ctypes.c_int64(Cin), H, W, Cin, C,
                      g(".adaptor.weight") if has_ad else None, g(".adaptor.bias") if has_ad else None,
                      g(".dc.0.weight"), g(".dc.0.bias"), g(".dc.2.weight"), g(".dc.2.bias"),
                      g(".dc.3.weight"), g(".dc.3.bias"), g(".ffn.0.weight"), g(".ffn.0.bias"),
                      g(".ffn.2.weight"), g(".ffn.2.bias"), int(shortcut), _ptr(qv), _ptr(out),
                      ctypes.c_int64(C))
        return out

    # ---- fp16-storage emulation of the HIP fp16 mode (tests only) -------------------------------------
    # The HIP fp16 kernels keep every activation in _Float16 and accumulate in fp32.
    # These two methods round to fp16 exactly where those kernels store (opendcvc_amd/csrc/dcvc_nn.hip:
    # dcb_head_kernel -> x', a;  dcb_tail_kernel -> d, W2 d + b2, o, v, r, out;  conv_kernel -> out) and
    # use the weights as dcvc_dcb_create / dcvc_conv_create pack them (pre-scaled by kAct, rounded to fp16),
    # so that the fp16 mode has a tight reference: what remains is the fp32 summation order inside the
    # MFMA and the hardware exp2 / rcp (about 1 ulp in fp32), i.e. rare one-ulp fp16 rounding flips.
    K_ACT = np.float32(-5.770780163555854)          # Traits<half_t>::kAct = -4 log2(e)

    @staticmethod
    def _rh(a):
        return np.asarray(a, np.float32).astype(np.float16).astype(np.float32)

    @classmethod
    def _gate16(cls, up):
        """Traits<half_t>::gate: g(u') = u' / (1 + 2^u') on the pre-scaled pre-activation (fp32)"""
        up = np.asarray(up, np.float32)
        with np.errstate(over="ignore"):
            return up * (np.float32(1.0) / (np.float32(1.0) + np.exp2(up)))

    def conv_f16(self, x, prefix, stride=1, pad=0, epilogue="bias", q=None):
        """conv_kernel in fp16 mode: fp16 weights / input, fp32 accumulate + bias, epilogue in fp32,
        one rounding to fp16 at the store.  epilogue: bias | quant (times q) | wsilu."""
        w = self._rh(self.sd[prefix + ".weight"])
        b = self.sd[prefix + ".bias"]
        x = self._rh(x)
        y = conv1x1(x, w, b) if (w.shape[2] == 1 and stride == 1) else conv2d(x, w, b, stride, pad)
        if epilogue == "quant":
            y = y * _f32(q).reshape(1, 1, -1)
        elif epilogue == "wsilu":
            with np.errstate(over="ignore"):
                y = y * (np.float32(1.0) / (np.float32(1.0) + np.exp2(y * self.K_ACT)))
        return self._rh(y)

    def dcb_f16(self, x, prefix, shortcut=False, q=None):
        """DepthConvBlock as dcb_head_kernel + dcb_tail_kernel compute it in fp16 mode."""
        sd, rh, ka = self.sd, self._rh, self.K_ACT
        g = lambda n: sd[prefix + n]
        C = g(".dc.0.weight").shape[0]
        x = rh(x)
        if (prefix + ".adaptor.weight") in sd:
            xi = rh(conv1x1(x, rh(g(".adaptor.weight")), g(".adaptor.bias")))          # x' (identity branch)
        else:
            xi = x
        a = rh(self._gate16(conv1x1(xi, rh(ka * g(".dc.0.weight")), ka * g(".dc.0.bias"))))   # kAct * wsilu(.)
        H, W, _ = a.shape
        d = np.empty((H, W, C), np.float32)
        wd = rh(g(".dc.2.weight") / ka)
        lib().orc_dw3x3(_ptr(_f32(a)), H, W, C, _ptr(_f32(wd)), _ptr(_f32(g(".dc.2.bias"))), _ptr(d))
        d = rh(d)
        o = rh(rh(conv1x1(d, rh(g(".dc.3.weight")), g(".dc.3.bias"))) + xi)
        u = conv1x1(o, rh(ka * g(".ffn.0.weight")), ka * g(".ffn.0.bias"))
        v = rh(self._gate16(u[:, :, :2 * C]) + self._gate16(u[:, :, 2 * C:]))
        r = rh(conv1x1(v, rh(g(".ffn.2.weight") / ka), g(".ffn.2.bias")) + o)
        if shortcut:
            r = r + xi
        if q is not None:
            r = r * _f32(q).reshape(1, 1, -1)
        return rh(r)

    def subpel(self, x, prefix, pad):
        """SubpelConv2x (layers.py:29-52): conv -> PixelShuffle(2)."""
        return pixel_shuffle(self.conv(x, prefix + ".conv.0", 1, pad), 2)

    def res_down(self, x, prefix):
        return self.dcb(self.conv(x, prefix + ".down", 2, 0), prefix + ".conv", shortcut=True)

    def res_up(self, x, prefix):
        return self.dcb(self.subpel(x, prefix + ".up", 0), prefix + ".conv", shortcut=True)


# --------------------------------------------------------------------------- entropy coding

class Coder:
    """EntropyCoder (entropy_models.py:11-81) over the C restatement of the rANS library."""

    def __init__(self):
        self.h = ctypes.c_void_p(lib().orc_coder_new())

    def __del__(self):
        try:
            lib().orc_coder_free(self.h)
        except Exception:
            pass

    def add_cdf(self, cdf, sizes, offsets):
        cdf = np.ascontiguousarray(cdf, np.int32)
        sizes = np.ascontiguousarray(sizes, np.int32)
        offsets = np.ascontiguousarray(offsets, np.int32)
        return lib().orc_add_cdf(self.h, _ptr(cdf), cdf.shape[0], cdf.shape[1], _ptr(sizes), _ptr(offsets))

    def set_use_two(self, two):
        lib().orc_enc_set_two(self.h, int(bool(two)))
        lib().orc_dec_set_two(self.h, int(bool(two)))

    def reset(self):
        lib().orc_enc_reset(self.h)

    def encode_y(self, packed, group):
        packed = np.ascontiguousarray(packed, np.int16)
        lib().orc_enc_y(self.h, _ptr(packed), packed.size, group)

    def encode_z(self, z, group, start, per_channel):
        z = np.ascontiguousarray(z, np.int8)
        lib().orc_enc_z(self.h, _ptr(z), z.size, group, start, per_channel)

    def flush(self):
        n = lib().orc_enc_flush(self.h)
        if n == 0:
            return b""
        buf = ctypes.cast(lib().orc_enc_stream(self.h), ctypes.POINTER(ctypes.c_uint8 * n)).contents
        return bytes(buf)

    def set_stream(self, data):
        a = np.frombuffer(bytes(data), np.uint8).copy()
        self._stream = a
        lib().orc_dec_set_stream(self.h, _ptr(a), a.size)

    def decode_y(self, idx, group):
        idx = np.ascontiguousarray(idx, np.uint8)
        out = np.empty(idx.size, np.int8)
        lib().orc_dec_y(self.h, _ptr(idx), idx.size, group, _ptr(out))
        return out

    def decode_z(self, total, group, start, per_channel):
        out = np.empty(total, np.int8)
        lib().orc_dec_z(self.h, total, group, start, per_channel, _ptr(out))
        return out


def pmf_to_quantized_cdf(pmf, precision=16):
    pmf = _f32(pmf)
    out = np.zeros(pmf.size + 1, np.uint32)
    r = lib().orc_pmf_to_quantized_cdf(_ptr(pmf), pmf.size, precision, _ptr(out))
    assert r == 0
    return out


def _pmf_to_cdf(pmf, tail_mass, pmf_length, max_length):
    """EntropyCoder.pmf_to_cdf (entropy_models.py:27-34)."""
    cdf = np.zeros((len(pmf_length), max_length + 2), np.int32)
    for i in range(len(pmf_length)):
        prob = np.concatenate([pmf[i, :pmf_length[i]], tail_mass[i]])
        c = pmf_to_quantized_cdf(prob, 16)
        cdf[i, :c.size] = c.astype(np.int32)
    return cdf


SCALE_MIN, SCALE_MAX, SCALE_LEVELS = 0.11, 16.0, 128
LOG_SCALE_MIN = math.log(SCALE_MIN)
LOG_STEP_RECIP = 1.0 / ((math.log(SCALE_MAX) - LOG_SCALE_MIN) / (SCALE_LEVELS - 1))


def gaussian_tables():
    """GaussianEncoder.update (entropy_models.py:244-283): 128 quantised Gaussian CDFs.  The fp32
    transcendental math is done with the same torch CPU operators the reference calls, because the
    integer tables define the bitstream."""
    import torch
    table = torch.exp(torch.linspace(math.log(SCALE_MIN), math.log(SCALE_MAX), SCALE_LEVELS))
    center = torch.zeros_like(table) + 8
    dist = torch.distributions.normal.Normal(0., torch.zeros_like(center) + table)
    for i in range(8, 1, -1):
        probs = torch.squeeze(dist.cdf(torch.zeros_like(center) + i))
        center = torch.where(probs > torch.zeros_like(center) + 0.9999, torch.zeros_like(center) + i, center)
    center = center.int()
    length = 2 * center + 1
    max_length = int(torch.max(length).item())
    samples = (torch.arange(max_length) - center[:, None]).float()
    dist = torch.distributions.normal.Normal(0., torch.zeros_like(samples) + table[:, None])
    upper = dist.cdf(samples + 0.5)
    lower = dist.cdf(samples - 0.5)
    pmf = (upper - lower).numpy()
    tail = (2 * lower[:, :1]).numpy()
    cdf = _pmf_to_cdf(pmf, tail, length.numpy(), max_length)
    return cdf, (length + 2).numpy().astype(np.int32), (-center).numpy().astype(np.int32)


def factorized_tables(sd, prefix, qp_num, channel):
    """BitEstimator.update (entropy_models.py:152-205)."""
    import torch
    import torch.nn.functional as F
    P = {k[len(prefix) + 1:]: torch.from_numpy(np.asarray(v, np.float32)) for k, v in sd.items()
         if k.startswith(prefix + ".")}

    def bitparm(x, f, final):
        x = x * F.softplus(P[f + ".h"]) + P[f + ".b"]
        if final:
            return x
        return x + torch.tanh(x) * torch.tanh(P[f + ".a"])

    def cdf_of(x):
        for f in ("f1", "f2", "f3"):
            x = bitparm(x, f, False)
        return torch.sigmoid(bitparm(x, "f4", True))

    medians = torch.zeros((qp_num, channel, 1, 1))
    minima = medians + 8
    for i in range(8, 1, -1):
        probs = cdf_of(torch.zeros_like(medians) - i)
        minima = torch.where(probs < torch.zeros_like(medians) + 0.0001, torch.zeros_like(medians) + i, minima)
    maxima = medians + 8
    for i in range(8, 1, -1):
        probs = cdf_of(torch.zeros_like(medians) + i)
        maxima = torch.where(probs > torch.zeros_like(medians) + 0.9999, torch.zeros_like(medians) + i, maxima)
    minima = minima.int()
    maxima = maxima.int()
    offset = -minima
    pmf_start = medians - minima
    pmf_length = maxima + minima + 1
    max_length = int(pmf_length.max())
    samples = torch.arange(max_length)[None, None, None, :] + pmf_start
    lower = cdf_of(samples - 0.5)
    upper = cdf_of(samples + 0.5)
    pmf = (upper - lower)[:, :, 0, :]
    upper = cdf_of(maxima.to(torch.float32))
    tail = lower[:, :, 0, :1] + (1.0 - upper[:, :, 0, -1:])
    cdf = _pmf_to_cdf(pmf.reshape(-1, max_length).numpy(), tail.reshape(-1, 1).numpy(),
                      pmf_length.reshape(-1).numpy(), max_length)
    return cdf, (pmf_length.reshape(-1) + 2).numpy().astype(np.int32), offset.reshape(-1).numpy().astype(np.int32)


# --------------------------------------------------------------------------- shared codec logic

def padding_size(h, w, p=64):
    """CompressionModel.get_padding_size (common_model.py:36-41) -> (pad_right, pad_bottom)."""
    return (w + p - 1) // p * p - w, (h + p - 1) // p * p - h


def downsampled_shape(h, w, p):
    """CompressionModel.get_downsampled_shape (common_model.py:44-47)."""
    nh = (h + p - 1) // p * p
    nw = (w + p - 1) // p * p
    return int(nh / p + 0.5), int(nw / p + 0.5)


def _parity_masks(H, W):
    hh, ww = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    return hh % 2, ww % 2


def masks_2x(H, W, C):
    """get_mask_2x (common_model.py:118-131), [H, W, C] float32."""
    hp, wp = _parity_masks(H, W)
    m0 = ((hp + wp) % 2 == 0).astype(np.float32)
    m1 = 1.0 - m0
    half = C // 2
    a = np.concatenate([np.repeat(m0[:, :, None], half, 2), np.repeat(m1[:, :, None], half, 2)], 2)
    b = np.concatenate([np.repeat(m1[:, :, None], half, 2), np.repeat(m0[:, :, None], half, 2)], 2)
    return _f32(a), _f32(b)


def masks_4x(H, W, C):
    """get_mask_4x (common_model.py:99-116)."""
    hp, wp = _parity_masks(H, W)
    m = [((hp == 0) & (wp == 0)), ((hp == 0) & (wp == 1)), ((hp == 1) & (wp == 0)), ((hp == 1) & (wp == 1))]
    m = [np.repeat(x.astype(np.float32)[:, :, None], C // 4, 2) for x in m]
    order = [(0, 1, 2, 3), (3, 2, 1, 0), (2, 3, 0, 1), (1, 0, 3, 2)]
    return [_f32(np.concatenate([m[i] for i in o], 2)) for o in order]


def process_with_mask(y, scales, means, mask, thres):
    """cuda_inference.py:58-74 (fallback branch), float32 throughout."""
    s_hat = scales * mask
    m_hat = means * mask
    y_res = (y - m_hat) * mask
    y_q = np.round(y_res)
    if thres is not None:
        y_q = y_q * (s_hat > np.float32(thres)).astype(np.float32)
    y_q = np.clip(y_q, np.float32(-128.), np.float32(127.))
    y_hat = y_q + m_hat
    return y_res, y_q, y_hat, s_hat


def _chw_flat(x):
    return np.ascontiguousarray(x.transpose(2, 0, 1)).reshape(-1)


class CodecBase:
    def __init__(self, sd, z_channel, qp_total):
        self.net = Net(sd)
        self.sd = self.net.sd
        self.z_channel = z_channel
        self.qp_total = qp_total
        self.coder = None
        self.thres = None

    def update(self, force_zero_thres=None):
        """CompressionModel.update (common_model.py:49-52)."""
        self.coder = Coder()
        self.thres = force_zero_thres
        self.g_tables = gaussian_tables()
        self.g_group = self.coder.add_cdf(*self.g_tables)
        self.z_tables = factorized_tables(self.sd, "bit_estimator_z", self.qp_total, self.z_channel)
        self.z_group = self.coder.add_cdf(*self.z_tables)

    def set_use_two_entropy_coders(self, two):
        self.coder.set_use_two(two)

    # -- gaussian symbol packing (entropy_models.py:288-341, cuda_inference.py:124-171)
    def _indexes(self, scales_flat):
        s = np.clip(scales_flat, np.float32(SCALE_MIN), np.float32(SCALE_MAX))
        idx = scale_to_index(s, SCALE_MIN, SCALE_MAX, LOG_SCALE_MIN, LOG_STEP_RECIP)
        keep = (s > np.float32(self.thres)) if self.thres is not None else None
        return idx, keep

    def pack_y(self, y_q_w, s_w):
        """[H,W,C'] symbols + scales -> compacted int16 stream in the reference's CHW order."""
        sym = _chw_flat(y_q_w).astype(np.int16)
        idx, keep = self._indexes(_chw_flat(s_w))
        packed = ((sym.astype(np.int32) << 8) + idx.astype(np.int32)).astype(np.int16)
        if keep is not None:
            packed = packed[keep]
        return packed

    def encode_y(self, y_q_w, s_w):
        packed = self.pack_y(y_q_w, s_w)
        self.coder.encode_y(packed, self.g_group)
        return packed

    def decode_y(self, scales_r):
        """decode_and_get_y (entropy_models.py:322-341): returns float32 [H,W,C'] symbols."""
        H, W, C = scales_r.shape
        idx, keep = self._indexes(_chw_flat(scales_r))
        if keep is not None:
            vals = self.coder.decode_y(idx[keep], self.g_group)
            out = np.zeros(idx.size, np.float32)
            out[keep] = vals.astype(np.float32)
        else:
            out = self.coder.decode_y(idx, self.g_group).astype(np.float32)
        return np.ascontiguousarray(out.reshape(C, H, W).transpose(1, 2, 0))

    def encode_z(self, z_hat, qp):
        H, W, C = z_hat.shape
        z8 = _chw_flat(z_hat).astype(np.int8)
        self.coder.encode_z(z8, self.z_group, qp * self.z_channel, H * W)
        return z8

    def decode_z(self, zh, zw, qp):
        v = self.coder.decode_z(self.z_channel * zh * zw, self.z_group, qp * self.z_channel, zh * zw)
        return np.ascontiguousarray(v.astype(np.float32).reshape(self.z_channel, zh, zw).transpose(1, 2, 0))

    def pad_for_y(self, y):
        H, W, _ = y.shape
        pr, pb = padding_size(H, W, 4)
        return replicate_pad(y, pb, pr)


# --------------------------------------------------------------------------- DMC (P frames)

class OracleDMC(CodecBase):
    QP_SHIFT = (0, 8, 4)

    def __init__(self, sd):
        super().__init__(sd, 128, 64 + 8)
        self.ref_feature = None     # [H/8, W/8, 256]
        self.ref_frame = None       # [H, W, 3]
        self.trace = {}

    # -- DPB (video_model.py:253-277,293-297)
    def clear_dpb(self):
        self.ref_feature = None
        self.ref_frame = None

    def add_ref_frame(self, feature=None, frame=None):
        self.ref_feature = feature
        self.ref_frame = nchw_to_hwc(frame) if frame is not None and frame.ndim == 4 else frame

    def reset_ref_feature(self):
        self.ref_feature = None

    def shift_qp(self, qp, fa_idx):
        return qp + self.QP_SHIFT[fa_idx]

    def prepare_feature_adaptor_i(self, last_qp):
        if self.ref_frame is None:
            self.ref_frame = np.clip(self.recon(self.ref_feature, self.q("q_recon", last_qp)), 0, 1)
            self.ref_feature = None

    def q(self, name, qp):
        return self.sd[name][qp, :, 0, 0]

    # -- sub-networks (forward_torch paths)
    def feature_adaptor(self):
        n = self.net
        if self.ref_feature is None:
            return n.dcb(pixel_unshuffle(self.ref_frame, 8), "feature_adaptor_i")
        return n.conv(self.ref_feature, "feature_adaptor_p")

    def extractor_part1(self, f, q_feature):
        n = self.net
        x1 = n.dcb(n.dcb(f, "feature_extractor.conv1.0"), "feature_extractor.conv1.1")
        return x1, x1 * q_feature

    def extractor_part2(self, x1):
        for i in range(4):
            x1 = self.net.dcb(x1, f"feature_extractor.conv2.{i}")
        return x1

    def encoder(self, x, ctx, q_enc):
        n = self.net
        f = n.conv(pixel_unshuffle(x, 8), "encoder.conv1")
        f = n.dcb(np.concatenate([f, ctx], 2), "encoder.conv2.0")
        f = n.dcb(f, "encoder.conv2.1")
        f = n.dcb(f, "encoder.conv3")
        f = f * q_enc
        return n.conv(f, "encoder.down", 2, 1)

    def hyper_encoder(self, y):
        n = self.net
        return n.res_down(n.res_down(n.dcb(y, "hyper_encoder.conv.0"), "hyper_encoder.conv.1"),
                          "hyper_encoder.conv.2")

    def prior_params(self, z_hat, ctx_t):
        n = self.net
        h = n.dcb(n.res_up(n.res_up(z_hat, "hyper_decoder.conv.0"), "hyper_decoder.conv.1"),
                  "hyper_decoder.conv.2")
        t = n.res_down(ctx_t, "temporal_prior_encoder")
        H, W, _ = t.shape
        p = np.concatenate([h[:H, :W], t], 2)
        for i in range(3):
            p = n.dcb(p, f"y_prior_fusion.conv.{i}")
        return n.conv(p, "y_prior_fusion.conv.3")

    def spatial_prior(self, x):
        n = self.net
        x = n.dcb(n.dcb(x, "y_spatial_prior.conv.0"), "y_spatial_prior.conv.1")
        return n.conv(x, "y_spatial_prior.conv.2")

    def decoder(self, y_hat, ctx, q_dec):
        n = self.net
        f = n.subpel(y_hat, "decoder.up", 1)
        f = n.dcb(np.concatenate([f, ctx], 2), "decoder.conv1.0")
        f = n.dcb(n.dcb(f, "decoder.conv1.1"), "decoder.conv1.2")
        return n.conv(f, "decoder.conv2") * q_dec

    def recon(self, feature, q_recon):
        n = self.net
        o = n.dcb(feature, "recon_generation_net.conv.0")
        for i in range(1, 4):
            o = n.dcb(o, f"recon_generation_net.conv.{i}")
        o = n.conv(o * q_recon, "recon_generation_net.head")
        return np.clip(pixel_shuffle(o, 8), np.float32(0), np.float32(1))

    # -- frame API
    def compress(self, x, qp):
        """video_model.py:299-341.  x: NCHW float [1,3,H,W] (padded to a multiple of 16)."""
        x = nchw_to_hwc(x)
        q_enc, q_dec, q_feat = self.q("q_encoder", qp), self.q("q_decoder", qp), self.q("q_feature", qp)
        f = self.feature_adaptor()
        x1, ctx_t = self.extractor_part1(f, q_feat)
        ctx = self.extractor_part2(x1)
        y = self.encoder(x, ctx, q_enc)
        z = self.hyper_encoder(self.pad_for_y(y))
        z_hat = np.clip(np.round(z), np.float32(-128), np.float32(127))
        params = self.prior_params(z_hat, ctx_t)
        # compress_prior_2x (common_model.py:143-161)
        C = y.shape[2]
        q_d = np.maximum(params[:, :, :C], np.float32(0.5))
        scales, means = params[:, :, C:2 * C], params[:, :, 2 * C:]
        yq = y * (np.float32(1.0) / q_d)
        m0, m1 = masks_2x(y.shape[0], y.shape[1], C)
        _, y_q_0, y_hat_0, s_hat_0 = process_with_mask(yq, scales, means, m0, self.thres)
        sp = self.spatial_prior(np.concatenate([y_hat_0, params], 2))
        scales1, means1 = sp[:, :, :C], sp[:, :, C:]
        _, y_q_1, y_hat_1, s_hat_1 = process_with_mask(yq, scales1, means1, m1, self.thres)
        y_hat = (y_hat_0 + y_hat_1) * q_d
        h = C // 2
        w = lambda a: a[:, :, :h] + a[:, :, h:]
        feature = self.decoder(y_hat, ctx, q_dec)
        self.coder.reset()
        z8 = self.encode_z(z_hat, qp)
        p0 = self.encode_y(w(y_q_0), w(s_hat_0))
        p1 = self.encode_y(w(y_q_1), w(s_hat_1))
        bits = self.coder.flush()
        self.trace = dict(y=y, z_hat=z_hat, params=params, y_hat=y_hat, feature=feature, ctx=ctx,
                          ctx_t=ctx_t, z8=z8, packed0=p0, packed1=p1)
        self.ref_feature, self.ref_frame = feature, None
        return {"bit_stream": bits}

    def decompress(self, bit_stream, sps, qp):
        """video_model.py:343-376."""
        q_dec, q_feat, q_rec = self.q("q_decoder", qp), self.q("q_feature", qp), self.q("q_recon", qp)
        self.coder.set_use_two(sps["ec_part"] == 1)
        self.coder.set_stream(bit_stream)
        zh, zw = downsampled_shape(sps["height"], sps["width"], 64)
        z_hat = self.decode_z(zh, zw, qp)
        f = self.feature_adaptor()
        x1, ctx_t = self.extractor_part1(f, q_feat)
        params = self.prior_params(z_hat, ctx_t)
        C = params.shape[2] // 3
        q_d = np.maximum(params[:, :, :C], np.float32(0.5))
        scales, means = params[:, :, C:2 * C], params[:, :, 2 * C:]
        H, W = means.shape[:2]
        m0, m1 = masks_2x(H, W, C)
        h = C // 2
        comb = lambda a, m: (a * m)[:, :, :h] + (a * m)[:, :, h:]
        y_q_r = self.decode_y(comb(scales, m0))
        y_hat_0 = (np.concatenate([y_q_r, y_q_r], 2) + means) * m0
        ctx = self.extractor_part2(x1)
        sp = self.spatial_prior(np.concatenate([y_hat_0, params], 2))
        scales1, means1 = sp[:, :, :C], sp[:, :, C:]
        y_q_r = self.decode_y(comb(scales1, m1))
        y_hat_1 = (np.concatenate([y_q_r, y_q_r], 2) + means1) * m1
        y_hat = (y_hat_0 + y_hat_1) * q_d
        feature = self.decoder(y_hat, ctx, q_dec)
        x_hat = self.recon(feature, q_rec)
        self.trace = dict(y_hat=y_hat, feature=feature, z_hat=z_hat)
        self.ref_feature, self.ref_frame = feature, x_hat
        return {"x_hat": hwc_to_nchw(x_hat)}


# --------------------------------------------------------------------------- DMCI (I frames)

class OracleDMCI(CodecBase):
    def __init__(self, sd):
        super().__init__(sd, 128, 64)
        self.trace = {}

    def q(self, name, qp):
        return self.sd[name][qp, :, 0, 0]

    def enc(self, x, q):
        n = self.net
        o = n.dcb(pixel_unshuffle(x, 8), "enc.enc_1", q=q)
        for i in range(6):
            o = n.dcb(o, f"enc.enc_2.{i}")
        return n.conv(o, "enc.enc_2.6", 2, 1)

    def dec(self, y_hat, q):
        n = self.net
        o = n.res_up(y_hat, "dec.dec_1.0")
        for i in range(1, 12):
            o = n.dcb(o, f"dec.dec_1.{i}")
        o = n.dcb(o, "dec.dec_1.12", q=q)
        o = n.dcb(o, "dec.dec_2")
        return np.clip(pixel_shuffle(o, 8), np.float32(0), np.float32(1))

    def hyper_enc(self, y):
        n = self.net
        return n.res_down(n.res_down(n.dcb(y, "hyper_enc.0"), "hyper_enc.1"), "hyper_enc.2")

    def prior_params(self, z_hat, yh, yw):
        n = self.net
        p = n.dcb(n.res_up(n.res_up(z_hat, "hyper_dec.0"), "hyper_dec.1"), "hyper_dec.2")
        for i in range(3):
            p = n.dcb(p, f"y_prior_fusion.{i}")
        p = n.conv(p, "y_prior_fusion.3")
        return np.ascontiguousarray(p[:yh, :yw])

    def spatial_prior(self, x, step):
        n = self.net
        x = n.dcb(x, f"y_spatial_prior_adaptor_{step}")
        for i in range(3):
            x = n.dcb(x, f"y_spatial_prior.{i}")
        return n.conv(x, "y_spatial_prior.3")

    def _separate(self, params):
        """separate_prior(is_video=False) (common_model.py:69-73)."""
        qq = sigmoid(params[:, :, :2]) * np.float32(1.5) + np.float32(0.5)
        C = (params.shape[2] - 2) // 2
        return qq[:, :, 0:1], qq[:, :, 1:2], params[:, :, 2:2 + C], params[:, :, 2 + C:]

    def compress(self, x, qp):
        """image_model.py:143-185 + compress_prior_4x (common_model.py:206-256)."""
        n = self.net
        x = nchw_to_hwc(x)
        y = self.enc(x, self.q("q_scale_enc", qp))
        z = self.hyper_enc(self.pad_for_y(y))
        z_hat = np.clip(np.round(z), np.float32(-128), np.float32(127))
        yh, yw, C = y.shape
        params = self.prior_params(z_hat, yh, yw)
        q_enc, q_dec, scales, means = self._separate(params)
        common = n.conv(params, "y_spatial_prior_reduction")
        masks = masks_4x(yh, yw, C)
        yq = y * q_enc
        w4 = lambda a: (a[:, :, :C // 4] + a[:, :, C // 4:C // 2]) + (a[:, :, C // 2:3 * C // 4] + a[:, :, 3 * C // 4:])
        sym, scl = [], []
        _, y_q, y_hat_k, s_hat = process_with_mask(yq, scales, means, masks[0], self.thres)
        sym.append(w4(y_q)); scl.append(w4(s_hat))
        so_far = y_hat_k
        for step in (1, 2, 3):
            sp = self.spatial_prior(np.concatenate([so_far, common], 2), step)
            scales, means = sp[:, :, :C], sp[:, :, C:]
            _, y_q, y_hat_k, s_hat = process_with_mask(yq, scales, means, masks[step], self.thres)
            sym.append(w4(y_q)); scl.append(w4(s_hat))
            so_far = so_far + y_hat_k
        y_hat = so_far * q_dec
        x_hat = self.dec(y_hat, self.q("q_scale_dec", qp))
        self.coder.reset()
        z8 = self.encode_z(z_hat, qp)
        packed = [self.encode_y(a, b) for a, b in zip(sym, scl)]
        bits = self.coder.flush()
        self.trace = dict(y=y, z_hat=z_hat, params=params, y_hat=y_hat, z8=z8, packed=packed)
        return {"bit_stream": bits, "x_hat": hwc_to_nchw(x_hat)}

    def decompress(self, bit_stream, sps, qp):
        """image_model.py:187-209 + decompress_prior_4x (common_model.py:258-296)."""
        n = self.net
        self.coder.set_use_two(sps["ec_part"] == 1)
        self.coder.set_stream(bit_stream)
        zh, zw = downsampled_shape(sps["height"], sps["width"], 64)
        yh, yw = downsampled_shape(sps["height"], sps["width"], 16)
        z_hat = self.decode_z(zh, zw, qp)
        params = self.prior_params(z_hat, yh, yw)
        _, q_dec, scales, means = self._separate(params)
        C = scales.shape[2]
        common = n.conv(params, "y_spatial_prior_reduction")
        masks = masks_4x(yh, yw, C)
        w4 = lambda a: (a[:, :, :C // 4] + a[:, :, C // 4:C // 2]) + (a[:, :, C // 2:3 * C // 4] + a[:, :, 3 * C // 4:])
        so_far = None
        for step in range(4):
            if step > 0:
                sp = self.spatial_prior(np.concatenate([so_far, common], 2), step)
                scales, means = sp[:, :, :C], sp[:, :, C:]
            y_q_r = self.decode_y(w4(scales * masks[step]))
            cur = (np.concatenate([y_q_r] * 4, 2) + means) * masks[step]
            so_far = cur if so_far is None else so_far + cur
        y_hat = so_far * q_dec
        x_hat = self.dec(y_hat, self.q("q_scale_dec", qp))
        self.trace = dict(y_hat=y_hat, z_hat=z_hat)
        return {"x_hat": hwc_to_nchw(x_hat)}
