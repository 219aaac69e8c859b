"""Host-side layer objects over the C ABI: they own packed device weights (created once) and
launch the HIP kernels on torch's current stream.  torch is only the container for device
memory; activations are HWC tensors [H, W, C_phys] (channel counts padded to a multiple of 32,
DepthConvBlock widths to a multiple of 64, pad channels are exact zeros).

Mirrors the reference's layer vocabulary (src/layers/layers.py): DepthConvBlock, SubpelConv2x,
ResidualBlockWithStride2, ResidualBlockUpsample, plus plain Conv2d.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import DcvcError, check

F16, F32 = _lib.F16, _lib.F32


def dtype_code(dt):
    if dt == torch.float16:
        return F16
    if dt == torch.float32:
        return F32
    raise DcvcError(f"unsupported dtype {dt}: the HIP path computes in float16 or float32")


def cpad(c, m=32):
    return (c + m - 1) // m * m


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _host(a):
    a = np.ascontiguousarray(a.detach().cpu().float().numpy() if isinstance(a, torch.Tensor) else a, dtype=np.float32)
    return a


def _hp(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def _geom(x):
    """(H, W, C, ld) of an HWC tensor/view usable by the kernels."""
    if x.dim() != 3 or x.stride(2) != 1 or x.stride(0) != x.shape[1] * x.stride(1):
        raise DcvcError(f"expected an HWC tensor with contiguous channels and uniform row stride, got "
                        f"shape {tuple(x.shape)} strides {tuple(x.stride())}")
    return x.shape[0], x.shape[1], x.shape[2], x.stride(1)


class Scratch:
    """Grow-only device scratch per (device, stream): no allocation in the steady state.  A buffer that has been
    handed out is never freed: captured HIP graphs (models.GraphCache) keep its raw address in their kernel
    arguments, so when a larger map needs a larger scratch the old one is retired, not released (sizes grow
    geometrically, the retired total stays below the largest buffer)."""
    _pool = {}
    _retired = []

    @classmethod
    def get(cls, nbytes, device):
        key = (str(device), torch.cuda.current_stream().cuda_stream)
        buf = cls._pool.get(key)
        if buf is None or buf.numel() < nbytes:
            if buf is not None:
                cls._retired.append(buf)
            buf = torch.empty(int(nbytes * 2) + 256, dtype=torch.uint8, device=device)
            cls._pool[key] = buf
        return buf


class DepthConvBlock:
    """Fused DepthConvBlock (reference: src/layers/layers.py:65-132, impl.cpp:7-121)."""

    def __init__(self, sd, prefix, dtype, shortcut=False):
        L = _lib.lib()
        g = lambda n: _host(sd[prefix + n])
        self.has_adaptor = (prefix + ".adaptor.weight") in sd
        w1 = g(".dc.0.weight")
        self.c = w1.shape[0]
        wa = g(".adaptor.weight") if self.has_adaptor else None
        ba = g(".adaptor.bias") if self.has_adaptor else None
        self.cin = wa.shape[1] if self.has_adaptor else self.c
        self.c_p = cpad(self.c, 64)
        self.cin_p = cpad(self.cin, 32) if self.has_adaptor else self.c_p
        self.dtype = dtype
        self.shortcut = bool(shortcut)
        arrs = [wa, ba, w1, g(".dc.0.bias"), g(".dc.2.weight"), g(".dc.2.bias"), g(".dc.3.weight"),
                g(".dc.3.bias"), g(".ffn.0.weight"), g(".ffn.0.bias"), g(".ffn.2.weight"), g(".ffn.2.bias")]
        h = ctypes.c_void_p()
        check(L.dcvc_dcb_create(dtype_code(dtype), self.cin, self.c, int(shortcut), *[_hp(a) for a in arrs],
                                ctypes.byref(h)), "dcvc_dcb_create")
        self.h = h

    def __del__(self):
        try:
            if getattr(self, "h", None):
                _lib.lib().dcvc_dcb_destroy(self.h)
        except Exception:
            pass

    def __call__(self, x0, x1=None, quant=None, out=None, head_done=False, a_slot=0, next_block=None):
        """x = concat(x0, x1) along channels; quant: float32 device tensor [C] or None;
        out: optional HWC view to write into (e.g. a slice of a concat buffer).
        head_done / a_slot / next_block: see dcb_chain() (dcvc_dcb_forward_chained)."""
        L = _lib.lib()
        H, W, c0, ld0 = _geom(x0)
        c1, ld1 = 0, 0
        if x1 is not None:
            H1, W1, c1, ld1 = _geom(x1)
            if (H1, W1) != (H, W):
                raise DcvcError("concat inputs differ in size")
        if out is None:
            out = torch.empty((H, W, self.c_p), dtype=self.dtype, device=x0.device)
        Ho, Wo, co, ldo = _geom(out)
        if (Ho, Wo) != (H, W) or co < self.c_p:
            raise DcvcError(f"output view {tuple(out.shape)} does not fit a {self.c_p}-channel block output")
        if x0.dtype != self.dtype or out.dtype != self.dtype:
            raise DcvcError("dtype mismatch between block and tensors")
        nbytes = L.dcvc_dcb_scratch_bytes(self.h, H, W)
        scratch = Scratch.get(nbytes, x0.device)
        check(L.dcvc_dcb_forward_chained(self.h, _p(x0), ld0, c0, _p(x1), ld1, c1, H, W, _p(quant), _p(out), ldo,
                                         _p(scratch), _stream(), int(head_done), int(a_slot),
                                         next_block.h if next_block is not None else None), "dcvc_dcb_forward")
        return out

    def then_conv(self, x0, x1, conv, conv_quant=None, out=None, head_done=False, a_slot=0):
        """This block followed by a 1x1 conv of the same width (dcvc_dcb_forward_then_conv): only the conv's output
        is written.  conv.fusable_after(self) must hold."""
        L = _lib.lib()
        H, W, c0, ld0 = _geom(x0)
        c1, ld1 = 0, 0
        if x1 is not None:
            H1, W1, c1, ld1 = _geom(x1)
            if (H1, W1) != (H, W):
                raise DcvcError("concat inputs differ in size")
        if out is None:
            out = torch.empty((H, W, conv.cout_p), dtype=self.dtype, device=x0.device)
        Ho, Wo, co, ldo = _geom(out)
        if (Ho, Wo) != (H, W) or co < conv.cout_p or out.dtype != self.dtype or x0.dtype != self.dtype:
            raise DcvcError(f"output view {tuple(out.shape)} does not fit the fused conv's output")
        scratch = Scratch.get(L.dcvc_dcb_scratch_bytes(self.h, H, W), x0.device)
        check(L.dcvc_dcb_forward_then_conv(self.h, _p(x0), ld0, c0, _p(x1), ld1, c1, H, W, _p(scratch), _stream(),
                                           int(head_done), int(a_slot), conv.h, _p(conv_quant), _p(out), ldo),
              "dcvc_dcb_forward_then_conv")
        return out

    def can_follow(self, prev, prev_quant):
        """True if `prev` (run with quant step `prev_quant`) may compute this block's first conv in its epilogue"""
        return (not self.has_adaptor and self.c_p == prev.c_p and self.dtype == prev.dtype and prev_quant is None
                and not prev.shortcut)


def dcb_chain(blocks, x0, x1=None, quant=None, out=None, return_all=False, then_conv=None, conv_quant=None):
    """A run of DepthConvBlocks feeding each other (nn.Sequential of DepthConvBlock in the reference models):
    wherever allowed, block i computes block i+1's pointwise first conv + activation on its output tile while
    that tile is still in LDS, so block i+1 starts at its depthwise stage (one launch and one activation read
    less per block; results are bit-identical to calling the blocks one by one).  x1 goes to the first
    block, quant / out to the last.  return_all: the list of every block's output instead of the last one (a run
    whose intermediate result is needed elsewhere too, e.g. the feature extractor's x1).
    then_conv: a Conv2d applied to the run's result (conv_quant: its quant vector; `out` then is the conv's output):
    where the shapes allow, the last block computes it on its output tile and the run's own result is never written."""
    x = x0
    fused = False
    outs = []
    for i, b in enumerate(blocks):
        last = i + 1 == len(blocks)
        if last and then_conv is not None and quant is None and not return_all and then_conv.fusable_after(b):
            return b.then_conv(x, x1 if i == 0 else None, then_conv, conv_quant, out=out, head_done=fused, a_slot=i & 1)
        qi = quant if last else None
        nxt = None if last or not blocks[i + 1].can_follow(b, qi) else blocks[i + 1]
        x = b(x, x1 if i == 0 else None, quant=qi, out=out if (last and then_conv is None) else None, head_done=fused, a_slot=i & 1,
              next_block=nxt)
        fused = nxt is not None
        outs.append(x)
    if then_conv is not None:
        return then_conv(x, quant=conv_quant, out=out)
    return outs if return_all else x


class Conv2d:
    """Dense conv (1x1 / 3x3 s1,s2 p1 / 2x2 s2) with fused epilogue; reference: nn.Conv2d uses in
    video_model.py / image_model.py / layers.py and the epilogue kernels of kernel.cu."""

    def __init__(self, sd, prefix, dtype, stride=1, pad=0, epilogue=_lib.EPI_BIAS):
        L = _lib.lib()
        w = _host(sd[prefix + ".weight"])
        b = _host(sd[prefix + ".bias"])
        self.cout, self.cin, self.kh, self.kw = w.shape
        self.stride, self.pad, self.epi, self.dtype = stride, pad, epilogue, dtype
        self.cin_p = cpad(self.cin, 32)
        if epilogue == _lib.EPI_SHUFFLE2:
            self.cout_p = cpad(self.cout // 4, 32)     # channels after PixelShuffle(2)
        else:
            self.cout_p = cpad(self.cout, 32)
        h = ctypes.c_void_p()
        check(L.dcvc_conv_create(dtype_code(dtype), self.cin, self.cout, self.kh, self.kw, stride, pad, epilogue,
                                 _hp(w), _hp(b), ctypes.byref(h)), "dcvc_conv_create")
        self.h = h

    def __del__(self):
        try:
            if getattr(self, "h", None):
                _lib.lib().dcvc_conv_destroy(self.h)
        except Exception:
            pass

    def fusable_after(self, block):
        """True if this conv may be computed in `block`'s tail (dcvc_dcb_forward_then_conv)"""
        return (self.kh == 1 and self.kw == 1 and self.stride == 1 and self.pad == 0 and self.cin == block.c and
                self.cin_p == block.c_p and self.cout_p == block.c_p and self.dtype == block.dtype and
                not block.shortcut and self.epi in (_lib.EPI_BIAS, _lib.EPI_BIAS_QUANT))

    def out_hw(self, H, W):
        Ho = (H + 2 * self.pad - self.kh) // self.stride + 1
        Wo = (W + 2 * self.pad - self.kw) // self.stride + 1
        if self.epi == _lib.EPI_SHUFFLE2:
            return 2 * Ho, 2 * Wo
        return Ho, Wo

    def __call__(self, x0, x1=None, quant=None, out=None, in_scale=None):
        """in_scale: float32 device vector [cin]: the conv of x * in_scale[c] (the product rounded to the element type,
        = scale_channels followed by the conv, without the intermediate tensor)"""
        L = _lib.lib()
        H, W, c0, ld0 = _geom(x0)
        c1, ld1 = 0, 0
        if x1 is not None:
            _, _, c1, ld1 = _geom(x1)
        if in_scale is not None and (in_scale.dtype != torch.float32 or in_scale.numel() < self.cin):
            raise DcvcError(f"in_scale must be a float32 vector of at least {self.cin} entries")
        Ho, Wo = self.out_hw(H, W)
        if out is None:
            out = torch.empty((Ho, Wo, self.cout_p), dtype=self.dtype, device=x0.device)
        H2, W2, co, ldo = _geom(out)
        if (H2, W2) != (Ho, Wo) or co < self.cout_p:
            raise DcvcError(f"output view {tuple(out.shape)} does not fit conv output {(Ho, Wo, self.cout_p)}")
        check(L.dcvc_conv_forward_scaled(self.h, _p(x0), ld0, c0, _p(x1), ld1, c1, H, W, _p(in_scale), _p(quant), _p(out),
                                         ldo, _stream()), "dcvc_conv_forward")
        return out


class SubpelConv2x:
    """conv + PixelShuffle(2) (reference: layers.py:29-62, SubpelConv2xProxy impl.cpp:123-167)."""

    def __init__(self, sd, prefix, dtype, pad):
        self.conv = Conv2d(sd, prefix + ".conv.0", dtype, 1, pad, _lib.EPI_SHUFFLE2)
        self.cout_p = self.conv.cout_p

    def __call__(self, x, out=None):
        return self.conv(x, out=out)


class ResidualBlockWithStride2:
    """reference: layers.py:135-144"""

    def __init__(self, sd, prefix, dtype):
        self.down = Conv2d(sd, prefix + ".down", dtype, 2, 0)
        self.conv = DepthConvBlock(sd, prefix + ".conv", dtype, shortcut=True)

    def __call__(self, x, out=None, in_scale=None):
        return self.conv(self.down(x, in_scale=in_scale), out=out)


class ResidualBlockUpsample:
    """reference: layers.py:147-156"""

    def __init__(self, sd, prefix, dtype):
        self.up = SubpelConv2x(sd, prefix + ".up", dtype, 0)
        self.conv = DepthConvBlock(sd, prefix + ".conv", dtype, shortcut=True)

    def __call__(self, x, out=None):
        return self.conv(self.up(x), out=out)


# ----------------------------------------------------------------------------- tensor helpers

def to_hwc(x_nchw, c_phys=None):
    """NCHW torch tensor [1,C,H,W] on the GPU -> HWC [H,W,c_phys] (zero padded), via the HIP kernel."""
    L = _lib.lib()
    _, C, H, W = x_nchw.shape
    cp_ = c_phys or cpad(C)
    x = x_nchw.contiguous()
    out = torch.zeros((H, W, cp_), dtype=x.dtype, device=x.device) if cp_ != C else \
        torch.empty((H, W, cp_), dtype=x.dtype, device=x.device)
    check(L.dcvc_nchw_to_hwc(dtype_code(x.dtype), _p(x), C, H * W, _p(out), cp_, _stream()), "dcvc_nchw_to_hwc")
    return out


def to_nchw(x_hwc, C=None):
    L = _lib.lib()
    H, W, cp_, ld = _geom(x_hwc)
    C = C or cp_
    out = torch.empty((1, C, H, W), dtype=x_hwc.dtype, device=x_hwc.device)
    check(L.dcvc_hwc_to_nchw(dtype_code(x_hwc.dtype), _p(x_hwc), ld, C, H * W, _p(out), _stream()), "dcvc_hwc_to_nchw")
    return out
