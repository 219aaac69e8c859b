"""Drop-in for the reference's operator module ``inference_extensions_cuda``
(src/layers/extensions/inference/bind.cpp:7-36, signatures def.h:6-107, import list
src/layers/cuda_inference.py:12-16): same function / class names, same argument order, torch
tensors in NCHW, B == 1, fp32 or fp16, outputs allocated by the callee unless an ``out`` argument
is given, in-place where the reference is in-place.  Registering this module under that name makes
the unmodified reference model code take its accelerated branch on ROCm (seam 2).

Every function launches HIP kernels of libdcvc_amd.so on torch's current stream; there is no
torch fallback.
"""
import ctypes

import torch

from . import _lib
from . import nn as L
from ._lib import DcvcError, check


def _s():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _req(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda or not t.is_contiguous():
            raise DcvcError("operator inputs must be contiguous CUDA tensors")
    return L.dtype_code(ts[0].dtype)


def process_with_mask_cuda(y, scales, means, mask, force_zero_thres):
    dt = _req(y, scales, means, mask)
    outs = [torch.empty_like(y) for _ in range(4)]
    check(_lib.lib().dcvc_op_process_with_mask(dt, L._p(y), L._p(scales), L._p(means), L._p(mask),
                                               float(force_zero_thres), *[L._p(o) for o in outs], y.numel(), _s()),
          "process_with_mask")
    return tuple(outs)   # y_res, y_q, y_hat, s_hat


def combine_for_reading_2x_cuda(out, x, mask):
    dt = _req(x, mask)
    check(_lib.lib().dcvc_op_combine_for_reading_2x(dt, L._p(out), L._p(x), L._p(mask), x.numel() // 2, _s()),
          "combine_for_reading_2x")


def restore_y_2x_cuda(out, y, means, mask):
    dt = _req(y, means, mask)
    if not out.is_contiguous():          # the reference passes a channel-slice of a wider buffer
        tmp = torch.empty_like(means)
        check(_lib.lib().dcvc_op_restore_y_2x(dt, L._p(tmp), L._p(y), L._p(means), L._p(mask), y.numel(), _s()),
              "restore_y_2x")
        out.copy_(tmp)
        return
    check(_lib.lib().dcvc_op_restore_y_2x(dt, L._p(out), L._p(y), L._p(means), L._p(mask), y.numel(), _s()),
          "restore_y_2x")


def restore_y_4x_cuda(out, y, means, mask):
    dt = _req(y, means, mask, out)
    check(_lib.lib().dcvc_op_restore_y_4x(dt, L._p(out), L._p(y), L._p(means), L._p(mask), y.numel(), _s()),
          "restore_y_4x")


def build_index_dec_cuda(out, cond_out, scales, scale_min, scale_max, log_scale_min, log_step_recip, skip_thres):
    dt = _req(scales)
    check(_lib.lib().dcvc_op_build_index_dec(dt, L._p(out), L._p(cond_out), L._p(scales), scale_min, scale_max,
                                             log_scale_min, log_step_recip, skip_thres, scales.numel(), _s()),
          "build_index_dec")


def build_index_enc_cuda(out, cond_out, symbols, scales, scale_min, scale_max, log_scale_min, log_step_recip,
                         skip_thres):
    dt = _req(scales, symbols)
    check(_lib.lib().dcvc_op_build_index_enc(dt, L._p(out), L._p(cond_out), L._p(symbols), L._p(scales), scale_min,
                                             scale_max, log_scale_min, log_step_recip, skip_thres, scales.numel(),
                                             _s()), "build_index_enc")


def round_and_to_int8_cuda(z):
    dt = _req(z)
    z8 = torch.empty(z.shape, dtype=torch.int8, device=z.device)
    check(_lib.lib().dcvc_op_round_and_to_int8(dt, L._p(z), L._p(z8), z.numel(), _s()), "round_and_to_int8")
    return z8


def clamp_reciprocal_with_quant_cuda(q_dec, y, min_val):
    dt = _req(y)
    q_dec = q_dec.contiguous()
    q_out = torch.empty_like(q_dec)
    check(_lib.lib().dcvc_op_clamp_reciprocal_with_quant(dt, L._p(q_dec), L._p(y), float(min_val), L._p(q_out),
                                                         y.numel(), _s()), "clamp_reciprocal_with_quant")
    return q_out


def add_and_multiply_cuda(x0, x1, q):
    dt = _req(x0, x1)
    q = q.contiguous()
    check(_lib.lib().dcvc_op_add_and_multiply(dt, L._p(x0), L._p(x1), L._p(q), x0.numel(), _s()), "add_and_multiply")


def bias_quant_cuda(x, bias, quant_step):
    dt = _req(x)
    _, C, H, W = x.shape
    check(_lib.lib().dcvc_op_bias_quant(dt, L._p(x), L._p(bias.contiguous()), L._p(quant_step.contiguous()), C, H * W,
                                        _s()), "bias_quant")


def bias_pixel_shuffle_8_cuda(out, x, bias, C, N, W, clamp):
    dt = _req(x, out)
    check(_lib.lib().dcvc_op_bias_pixel_shuffle_8(dt, L._p(out), L._p(x), L._p(bias.contiguous()), C, N // W, W,
                                                  int(bool(clamp)), _s()), "bias_pixel_shuffle_8")


def replicate_pad_cuda(x, padB, padR):
    dt = L.dtype_code(x.dtype) if x.dtype in (torch.float16, torch.float32) else None
    if dt is None:
        raise DcvcError("replicate_pad: float16/float32 only")
    x = x.contiguous()
    B, C, H, W = x.shape
    out = torch.empty((B, C, H + padB, W + padR), dtype=x.dtype, device=x.device)
    check(_lib.lib().dcvc_op_replicate_pad(dt, L._p(x), B * C, H, W, padB, padR, L._p(out), _s()), "replicate_pad")
    return out


def bias_wsilu_depthwise_conv2d_cuda(x, weight, bias):
    dt = _req(x)
    _, C, H, W = x.shape
    out = torch.empty_like(x)
    check(_lib.lib().dcvc_op_bias_wsilu_depthwise_conv2d(dt, L._p(x), L._p(weight.contiguous()),
                                                         L._p(bias.contiguous()), C, H, W, L._p(out), _s()),
          "bias_wsilu_depthwise_conv2d")
    return out


class DepthConvProxy:
    """reference: DepthConvProxy (def.h:53-91, impl.cpp:7-121) - NCHW in / out around the fused block."""

    def __init__(self):
        self._blk = None
        self._c = None

    def _make(self, names, tensors, shortcut):
        sd = {"m." + n: t for n, t in zip(names, tensors)}
        self._blk = L.DepthConvBlock(sd, "m", tensors[0].dtype, shortcut=bool(shortcut))
        self._c = self._blk.c

    def set_param(self, dc_conv1_weight, dc_conv1_bias, dc_depth_conv_weight, dc_depth_conv_bias, dc_conv2_weight,
                  dc_conv2_bias, ffn_conv1_weight, ffn_conv1_bias, ffn_conv2_weight, ffn_conv2_bias, shortcut):
        names = ["dc.0.weight", "dc.0.bias", "dc.2.weight", "dc.2.bias", "dc.3.weight", "dc.3.bias", "ffn.0.weight",
                 "ffn.0.bias", "ffn.2.weight", "ffn.2.bias"]
        self._make(names, [dc_conv1_weight, dc_conv1_bias, dc_depth_conv_weight, dc_depth_conv_bias, dc_conv2_weight,
                           dc_conv2_bias, ffn_conv1_weight, ffn_conv1_bias, ffn_conv2_weight, ffn_conv2_bias], shortcut)

    def set_param_with_adaptor(self, dc_conv1_weight, dc_conv1_bias, dc_depth_conv_weight, dc_depth_conv_bias,
                               dc_conv2_weight, dc_conv2_bias, ffn_conv1_weight, ffn_conv1_bias, ffn_conv2_weight,
                               ffn_conv2_bias, adaptor_weight, adaptor_bias, shortcut):
        names = ["dc.0.weight", "dc.0.bias", "dc.2.weight", "dc.2.bias", "dc.3.weight", "dc.3.bias", "ffn.0.weight",
                 "ffn.0.bias", "ffn.2.weight", "ffn.2.bias", "adaptor.weight", "adaptor.bias"]
        self._make(names, [dc_conv1_weight, dc_conv1_bias, dc_depth_conv_weight, dc_depth_conv_bias, dc_conv2_weight,
                           dc_conv2_bias, ffn_conv1_weight, ffn_conv1_bias, ffn_conv2_weight, ffn_conv2_bias,
                           adaptor_weight, adaptor_bias], shortcut)

    def _run(self, x, quant=None):
        if self._blk is None:
            raise DcvcError("DepthConvProxy: set_param has not been called")
        q = quant.reshape(-1).float().contiguous() if quant is not None else None
        return L.to_nchw(self._blk(L.to_hwc(x, self._blk.cin_p), quant=q), self._c)

    def forward(self, x):
        return self._run(x)

    def forward_with_quant_step(self, x, quant_step):
        return self._run(x, quant_step)

    def forward_with_cat(self, x, to_cat, cat_at_front):
        out = self._run(x)
        return torch.cat((to_cat, out), dim=1) if cat_at_front else torch.cat((out, to_cat), dim=1)


class SubpelConv2xProxy:
    """reference: SubpelConv2xProxy (def.h:93-107, impl.cpp:123-167)"""

    def __init__(self):
        self._conv = None

    def set_param(self, weight, bias, padding):
        self._conv = L.Conv2d({"m.weight": weight, "m.bias": bias}, "m", weight.dtype, 1, int(padding),
                              _lib.EPI_SHUFFLE2)

    def forward(self, x):
        if self._conv is None:
            raise DcvcError("SubpelConv2xProxy: set_param has not been called")
        return L.to_nchw(self._conv(L.to_hwc(x, self._conv.cin_p)), self._conv.cout // 4)

    def forward_with_cat(self, x, to_cat, cat_at_front):
        out = self.forward(x)
        return torch.cat((to_cat, out), dim=1) if cat_at_front else torch.cat((out, to_cat), dim=1)
