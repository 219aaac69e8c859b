"""DMCI (intra) and DMC (inter) frame codecs on the HIP path - drop-in for the reference's
``src.models.image_model.DMCI`` / ``src.models.video_model.DMC`` as driven by test_video.py
(SURVEY.md section 8b seam 1): same constructor, ``load_state_dict`` of a reference checkpoint,
``.to(device) / .eval() / .update(force_zero_thres) / .half()``, ``set_use_two_entropy_coders``,
``compress(x, qp)`` / ``decompress(bit_stream, sps, qp)`` and the DPB helpers.

torch holds the parameters and device buffers; every per-frame computation is a HIP kernel
launched through the C ABI (opendcvc_amd/nn.py, _lib.py) and the entropy coding runs in the C++
host coder.  There is no torch fallback: without a GPU or without libdcvc_amd.so these classes
raise DcvcError.

Reference behaviour restated (no code shared): video_model.py:226-379, image_model.py:102-209,
common_model.py:13-296.
"""
import ctypes

import numpy as np
import torch
from torch import nn as tnn

from . import _lib, arch, entropy
from . import nn as L
from ._lib import DcvcError, check


def _register(root, dotted, tensor):
    parts = dotted.split(".")
    m = root
    for p in parts[:-1]:
        if not hasattr(m, p):
            m.add_module(p, tnn.Module())
        m = getattr(m, p)
    m.register_parameter(parts[-1], tnn.Parameter(tensor, requires_grad=False))


class RefFrame:
    def __init__(self):
        self.frame = None      # NCHW reconstruction (what the reference stores)
        self.feature = None    # HWC feature on the HIP path
        self.poc = None


class CompressionModel(tnn.Module):
    """reference: CompressionModel (common_model.py:13-61)"""

    def __init__(self, model_name, z_channel, qp_total):
        super().__init__()
        self._model_name = model_name
        self.z_channel = z_channel
        self.qp_total = qp_total
        for name, shape, kind in arch.spec_for(model_name).items:
            init = torch.ones(shape) if kind == "q" else torch.zeros(shape)
            _register(self, name, init)
        self.entropy_coder = None
        self.force_zero_thres = None
        self._layers = None
        self._layers_key = None
        self._q = {}

    # ---- static helpers used by test_video.py
    @staticmethod
    def get_qp_num():
        return arch.QP_NUM

    @staticmethod
    def get_padding_size(height, width, p=64):
        new_h = (height + p - 1) // p * p
        new_w = (width + p - 1) // p * p
        return new_w - width, new_h - height

    @staticmethod
    def get_downsampled_shape(height, width, p):
        new_h = (height + p - 1) // p * p
        new_w = (width + p - 1) // p * p
        return int(new_h / p + 0.5), int(new_w / p + 0.5)

    def update(self, force_zero_thres=None):
        """Builds the entropy coder and its CDF tables (common_model.py:49-52)."""
        _lib.lib()
        self.force_zero_thres = force_zero_thres
        self.entropy_coder = entropy.EntropyCoder()
        self._g_group = self.entropy_coder.add_cdf(*entropy.gaussian_cdf_tables())
        sd = self.state_dict()
        pre = "bit_estimator_z."
        params = {k[len(pre):]: v.detach().float().cpu() for k, v in sd.items() if k.startswith(pre)}
        self._z_group = self.entropy_coder.add_cdf(*entropy.factorized_cdf_tables(params, self.qp_total, self.z_channel))

    def set_use_two_entropy_coders(self, use_two_entropy_coders):
        self.entropy_coder.set_use_two_entropy_coders(use_two_entropy_coders)

    # ---- plumbing
    def _dtype_device(self):
        p = next(self.parameters())
        return p.dtype, p.device

    def _ensure_layers(self):
        dtype, device = self._dtype_device()
        if device.type != "cuda":
            raise DcvcError("the DCVC-RT hot path runs on an MI355X only: move the model to a cuda device "
                            "(there is no CPU fallback)")
        if self.entropy_coder is None:
            raise DcvcError("call update(force_zero_thres) before compress/decompress")
        key = (dtype, str(device))
        if self._layers_key != key:
            _lib.require_gpu()
            torch.cuda.set_device(device)
            sd = {k: v.detach().float().cpu() for k, v in self.state_dict().items()}
            self._layers = self._build_layers(sd, dtype)
            self._q = {k: v.reshape(v.shape[0], v.shape[1]).to(device=device, dtype=torch.float32).contiguous()
                       for k, v in sd.items() if k.startswith("q_")}
            self._layers_key = key
        return dtype, device

    def _thres(self):
        return -1.0 if self.force_zero_thres is None else float(self.force_zero_thres)

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def _pad_for_y(self, y):
        H, W, C = y.shape
        pr, pb = self.get_padding_size(H, W, 4)
        if pr == 0 and pb == 0:
            return y
        out = torch.empty((H + pb, W + pr, C), dtype=y.dtype, device=y.device)
        check(_lib.lib().dcvc_replicate_pad_hwc(L.dtype_code(y.dtype), L._p(y), C, H, W, C, pb, pr, L._p(out), C,
                                                self._stream()), "replicate_pad")
        return out

    def _unshuffle8(self, x):
        _, C, H, W = x.shape
        x = x.contiguous()
        out = torch.empty((H // 8, W // 8, C * 64), dtype=x.dtype, device=x.device)
        check(_lib.lib().dcvc_unshuffle8(L.dtype_code(x.dtype), L._p(x), C, H, W, L._p(out), C * 64, self._stream()),
              "unshuffle8")
        return out

    def _shuffle8_clamp(self, x, bias=None):
        H, W, C, ld = L._geom(x)
        out = torch.empty((1, 3, H * 8, W * 8), dtype=x.dtype, device=x.device)
        check(_lib.lib().dcvc_shuffle8_clamp(L.dtype_code(x.dtype), L._p(x), ld, L._p(bias), 3, H, W, 1, L._p(out),
                                             self._stream()), "shuffle8_clamp")
        return out

    def _crop(self, x, H2, W2, out=None):
        H, W, C, ld = L._geom(x)
        if out is None:
            if (H, W) == (H2, W2):
                return x
            out = torch.empty((H2, W2, C), dtype=x.dtype, device=x.device)
        _, _, _, ldo = L._geom(out)
        check(_lib.lib().dcvc_crop_hwc(L.dtype_code(x.dtype), L._p(x), ld, W, H2, W2, C, L._p(out), ldo,
                                       self._stream()), "crop")
        return out

    # ---- z
    def _quantize_z(self, z):
        H, W, C, ld = L._geom(z)
        z8 = torch.empty(C * H * W, dtype=torch.int8, device=z.device)
        check(_lib.lib().dcvc_round_z(L.dtype_code(z.dtype), L._p(z), ld, H, W, C, L._p(z8), self._stream()), "round_z")
        return z, z8

    def _z_to_device(self, z_host, zh, zw, dtype, device):
        z8 = torch.empty(self.z_channel * zh * zw, dtype=torch.int8, device=device)
        check(_lib.lib().dcvc_memcpy_h2d(L._p(z8), z_host.ctypes.data_as(ctypes.c_void_p), z8.numel(), self._stream()), "h2d")
        out = torch.empty((zh, zw, self.z_channel), dtype=dtype, device=device)
        check(_lib.lib().dcvc_z_from_int8(L.dtype_code(dtype), L._p(z8), zh, zw, self.z_channel, L._p(out),
                                          self.z_channel, self._stream()), "z_from_int8")
        return out

    # ---- host <-> device staging
    def _d2h(self, key, t):
        buf = self.entropy_coder.pinned(key, t.numel() * t.element_size())
        check(_lib.lib().dcvc_memcpy_d2h(ctypes.c_void_p(buf.ptr), L._p(t), t.numel() * t.element_size(), self._stream()), "d2h")
        return buf

    def _prior_enc_step(self, groups, step, q_mode, y, qsrc, scales, means, yhat, packed):
        H, W, C, ldy = L._geom(y)
        check(_lib.lib().dcvc_prior_enc_step(
            L.dtype_code(y.dtype), groups, step, q_mode, L._p(y), ldy, L._p(qsrc), qsrc.stride(1), L._p(scales),
            scales.stride(1), L._p(means), means.stride(1), H, W, C, self._thres(), L._p(yhat), yhat.stride(1),
            L._p(yhat), yhat.stride(1), L._p(packed), self._stream()), "prior_enc_step")

    def _prior_dec_index(self, groups, step, scales, H, W, C, idx):
        check(_lib.lib().dcvc_prior_dec_index(L.dtype_code(scales.dtype), groups, step, L._p(scales), scales.stride(1),
                                              H, W, C, self._thres(), L._p(idx), self._stream()), "prior_dec_index")

    def _prior_dec_restore(self, groups, step, sym, means, yhat, H, W, C):
        check(_lib.lib().dcvc_prior_dec_restore(L.dtype_code(means.dtype), groups, step, L._p(sym), L._p(means),
                                                means.stride(1), H, W, C, L._p(yhat), yhat.stride(1), L._p(yhat),
                                                yhat.stride(1), self._stream()), "prior_dec_restore")

    def _prior_finish(self, q_mode, yhat, qsrc):
        H, W, C, ld = L._geom(yhat)
        check(_lib.lib().dcvc_prior_finish(L.dtype_code(yhat.dtype), q_mode, L._p(yhat), ld, L._p(qsrc), qsrc.stride(1),
                                           H, W, C, self._stream()), "prior_finish")

    def _decode_step_begin(self, groups, step, scales, H, W, C, key):
        """first half of a checkerboard decode step: cdf indexes on the GPU -> pinned host buffer.
        Returns a ticket; GPU work queued after this call overlaps with the host entropy decoding."""
        n = (C // groups) * H * W
        idx = torch.empty(n, dtype=torch.uint8, device=scales.device)
        self._prior_dec_index(groups, step, scales, H, W, C, idx)
        hb = self._d2h(key + "_idx", idx)
        ev = torch.cuda.Event()
        ev.record()
        return (n, hb, ev, key, idx)

    def _decode_step_end(self, ticket, groups, step, means, yhat, H, W, C):
        """second half: wait for the indexes only, rANS-decode on the host, upload, restore y_hat."""
        n, hb, ev, key, _ = ticket
        ev.synchronize()
        sb = self.entropy_coder.pinned(key + "_sym", n)
        self.entropy_coder.decode_and_get_y(hb.view(np.uint8, n), self._g_group, sb.view(np.int8, n))
        sym = torch.empty(n, dtype=torch.int8, device=yhat.device)
        check(_lib.lib().dcvc_memcpy_h2d(L._p(sym), ctypes.c_void_p(sb.ptr), n, self._stream()), "h2d")
        self._prior_dec_restore(groups, step, sym, means, yhat, H, W, C)

    def _decode_step(self, groups, step, scales, means, yhat, H, W, C, key):
        self._decode_step_end(self._decode_step_begin(groups, step, scales, H, W, C, key), groups, step, means, yhat,
                              H, W, C)


# =============================================================================== DMC (P frames)

class DMC(CompressionModel):
    """reference: DMC (video_model.py:226-379)"""

    def __init__(self):
        super().__init__("dmc", arch.DMC_CH_Z, arch.QP_NUM + arch.DMC_EXTRA_QP)
        self.qp_shift = list(arch.DMC_QP_SHIFT)
        self.dpb = []
        self.max_dpb_size = 1
        self.curr_poc = 0

    def _build_layers(self, sd, dt):
        D, C2, R = L.DepthConvBlock, L.Conv2d, L.ResidualBlockWithStride2
        n = {}
        n["fa_i"] = D(sd, "feature_adaptor_i", dt)
        n["fa_p"] = C2(sd, "feature_adaptor_p", dt)
        n["fe1"] = [D(sd, f"feature_extractor.conv1.{i}", dt) for i in range(2)]
        n["fe2"] = [D(sd, f"feature_extractor.conv2.{i}", dt) for i in range(4)]
        n["enc_conv1"] = C2(sd, "encoder.conv1", dt)
        n["enc_conv2"] = [D(sd, f"encoder.conv2.{i}", dt) for i in range(2)]
        n["enc_conv3"] = D(sd, "encoder.conv3", dt)
        n["enc_down"] = C2(sd, "encoder.down", dt, 2, 1)
        n["hyper_enc"] = [D(sd, "hyper_encoder.conv.0", dt), R(sd, "hyper_encoder.conv.1", dt), R(sd, "hyper_encoder.conv.2", dt)]
        n["hyper_dec"] = [L.ResidualBlockUpsample(sd, "hyper_decoder.conv.0", dt),
                          L.ResidualBlockUpsample(sd, "hyper_decoder.conv.1", dt), D(sd, "hyper_decoder.conv.2", dt)]
        n["temporal"] = R(sd, "temporal_prior_encoder", dt)
        n["fusion"] = [D(sd, f"y_prior_fusion.conv.{i}", dt) for i in range(3)]
        n["fusion_out"] = C2(sd, "y_prior_fusion.conv.3", dt)
        n["spatial"] = [D(sd, "y_spatial_prior.conv.0", dt), D(sd, "y_spatial_prior.conv.1", dt)]
        n["spatial_out"] = C2(sd, "y_spatial_prior.conv.2", dt)
        n["dec_up"] = L.SubpelConv2x(sd, "decoder.up", dt, 1)
        n["dec_conv1"] = [D(sd, f"decoder.conv1.{i}", dt) for i in range(3)]
        n["dec_conv2"] = C2(sd, "decoder.conv2", dt, epilogue=_lib.EPI_BIAS_QUANT)
        n["recon"] = [D(sd, f"recon_generation_net.conv.{i}", dt) for i in range(4)]
        n["recon_head"] = C2(sd, "recon_generation_net.head", dt)
        return n

    # ---- DPB (video_model.py:253-277)
    def reset_ref_feature(self):
        if len(self.dpb) > 0:
            self.dpb[0].feature = None

    def add_ref_frame(self, feature=None, frame=None, increase_poc=True):
        ref = RefFrame()
        ref.poc = self.curr_poc
        ref.frame = frame
        ref.feature = feature
        if len(self.dpb) >= self.max_dpb_size:
            self.dpb.pop(-1)
        self.dpb.insert(0, ref)
        if increase_poc:
            self.curr_poc += 1

    def clear_dpb(self):
        self.dpb.clear()

    def set_curr_poc(self, poc):
        self.curr_poc = poc

    def shift_qp(self, qp, fa_idx):
        return qp + self.qp_shift[fa_idx]

    def prepare_feature_adaptor_i(self, last_qp):
        if self.dpb[0].frame is None:
            self._ensure_layers()
            self.dpb[0].frame = self._recon(self.dpb[0].feature, self._q["q_recon"][last_qp])
            self.reset_ref_feature()

    # ---- sub-networks
    def _apply_feature_adaptor(self):
        n = self._layers
        ref = self.dpb[0]
        if ref.feature is None:
            dtype, _ = self._dtype_device()
            return n["fa_i"](self._unshuffle8(ref.frame.to(dtype)))
        return n["fa_p"](ref.feature)

    def _extractor_part1(self, f, q_feature):
        n = self._layers
        x1 = n["fe1"][1](n["fe1"][0](f))
        ctx_t = torch.empty_like(x1)
        H, W, C, ld = L._geom(x1)
        check(_lib.lib().dcvc_scale_channels(L.dtype_code(x1.dtype), L._p(x1), ld, L._p(q_feature), H * W, C,
                                             L._p(ctx_t), C, self._stream()), "scale_channels")
        return x1, ctx_t

    def _extractor_part2(self, x1):
        for blk in self._layers["fe2"]:
            x1 = blk(x1)
        return x1

    def _prior_params(self, z_hat, ctx_t, yh, yw):
        """res_prior_param_decoder (video_model.py:279-286) -> [yh, yw, 384] = q_dec | scales | means"""
        n = self._layers
        cat = torch.empty((yh, yw, 3 * arch.DMC_CH_Y), dtype=z_hat.dtype, device=z_hat.device)
        h = n["hyper_dec"][1](n["hyper_dec"][0](z_hat))
        if h.shape[0] == yh and h.shape[1] == yw:
            n["hyper_dec"][2](h, out=cat[:, :, :arch.DMC_CH_Y])
        else:
            self._crop(n["hyper_dec"][2](h), yh, yw, out=cat[:, :, :arch.DMC_CH_Y])
        n["temporal"](ctx_t, out=cat[:, :, arch.DMC_CH_Y:])
        p = cat
        for blk in n["fusion"]:
            p = blk(p)
        return n["fusion_out"](p)

    def _spatial_prior(self, y_hat, params):
        n = self._layers
        return n["spatial_out"](n["spatial"][1](n["spatial"][0](y_hat, params)))

    def _decoder(self, y_hat, ctx, q_decoder):
        n = self._layers
        f = n["dec_conv1"][0](n["dec_up"](y_hat), ctx)
        f = n["dec_conv1"][2](n["dec_conv1"][1](f))
        return n["dec_conv2"](f, quant=q_decoder)

    def _recon(self, feature, q_recon):
        n = self._layers
        o = n["recon"][2](n["recon"][1](n["recon"][0](feature)))
        o = n["recon"][3](o, quant=q_recon)
        return self._shuffle8_clamp(n["recon_head"](o))

    # ---- frame API
    def compress(self, x, qp):
        """video_model.py:299-341.  x: [1,3,H,W] in [0,1], H and W multiples of 16."""
        dtype, device = self._ensure_layers()
        n = self._layers
        C = arch.DMC_CH_Y
        x = x.to(device=device, dtype=dtype)
        q_enc, q_dec, q_feat = self._q["q_encoder"][qp], self._q["q_decoder"][qp], self._q["q_feature"][qp]

        f = self._apply_feature_adaptor()
        x1, ctx_t = self._extractor_part1(f, q_feat)
        ctx = self._extractor_part2(x1)
        e = n["enc_conv2"][0](n["enc_conv1"](self._unshuffle8(x)), ctx)
        e = n["enc_conv3"](n["enc_conv2"][1](e), quant=q_enc)
        y = n["enc_down"](e)
        yh, yw = y.shape[0], y.shape[1]

        z = n["hyper_enc"][2](n["hyper_enc"][1](n["hyper_enc"][0](self._pad_for_y(y))))
        z_hat, z8 = self._quantize_z(z)
        params = self._prior_params(z_hat, ctx_t, yh, yw)

        nsym = (C // 2) * yh * yw
        y_hat = torch.empty((yh, yw, C), dtype=dtype, device=device)
        packed = torch.empty((2, nsym), dtype=torch.int16, device=device)
        self._prior_enc_step(2, 0, 0, y, params[:, :, :C], params[:, :, C:2 * C], params[:, :, 2 * C:], y_hat, packed[0])
        sp = self._spatial_prior(y_hat, params)
        self._prior_enc_step(2, 1, 0, y, params[:, :, :C], sp[:, :, :C], sp[:, :, C:], y_hat, packed[1])
        self._prior_finish(0, y_hat, params[:, :, :C])

        hz = self._d2h("z8", z8)
        hp = self._d2h("packed", packed)
        ready = torch.cuda.Event()
        ready.record()
        feature = self._decoder(y_hat, ctx, q_dec)     # GPU keeps going while the host codes

        ready.synchronize()
        ec = self.entropy_coder
        ec.reset()
        zh, zw = z.shape[0], z.shape[1]
        ec.encode_z(hz.view(np.int8, z8.numel()), self._z_group, qp * self.z_channel, zh * zw)
        ps = hp.view(np.int16, 2 * nsym)
        ec.encode_y(ps[:nsym], self._g_group, borrowed=True)     # pinned staging buffer, untouched until
        ec.encode_y(ps[nsym:], self._g_group, borrowed=True)     # get_encoded_stream() below
        ec.flush()
        bit_stream = ec.get_encoded_stream()
        # no device synchronisation here (the reference has none either): the tail of the decoder stays in
        # flight on this stream and overlaps the caller's next host work; callers that time a frame sync.
        self.add_ref_frame(feature, None)
        return {"bit_stream": bit_stream}

    def decompress(self, bit_stream, sps, qp):
        """video_model.py:343-376"""
        dtype, device = self._ensure_layers()
        C = arch.DMC_CH_Y
        q_dec, q_feat, q_rec = self._q["q_decoder"][qp], self._q["q_feature"][qp], self._q["q_recon"][qp]
        ec = self.entropy_coder
        ec.set_use_two_entropy_coders(sps["ec_part"] == 1)
        ec.set_stream(bit_stream)
        zh, zw = self.get_downsampled_shape(sps["height"], sps["width"], 64)
        yh, yw = self.get_downsampled_shape(sps["height"], sps["width"], 16)
        ec.decode_z(self.z_channel * zh * zw, self._z_group, qp * self.z_channel, zh * zw)

        f = self._apply_feature_adaptor()
        x1, ctx_t = self._extractor_part1(f, q_feat)

        zb = ec.pinned("z_dec", self.z_channel * zh * zw)
        ec.get_decoded(zb.view(np.int8, self.z_channel * zh * zw))
        z_hat = self._z_to_device(zb.view(np.int8, self.z_channel * zh * zw), zh, zw, dtype, device)
        params = self._prior_params(z_hat, ctx_t, yh, yw)
        y_hat = torch.empty((yh, yw, C), dtype=dtype, device=device)
        ticket = self._decode_step_begin(2, 0, params[:, :, C:2 * C], yh, yw, C, "p0")
        ctx = self._extractor_part2(x1)          # runs on the GPU while the host decodes step 0
        self._decode_step_end(ticket, 2, 0, params[:, :, 2 * C:], y_hat, yh, yw, C)
        sp = self._spatial_prior(y_hat, params)
        self._decode_step(2, 1, sp[:, :, :C], sp[:, :, C:], y_hat, yh, yw, C, "p1")
        self._prior_finish(0, y_hat, params[:, :, :C])

        feature = self._decoder(y_hat, ctx, q_dec)
        x_hat = self._recon(feature, q_rec)
        self.add_ref_frame(feature, x_hat)
        return {"x_hat": x_hat}


# =============================================================================== DMCI (I frames)

class DMCI(CompressionModel):
    """reference: DMCI (image_model.py:102-209)"""

    def __init__(self, N=arch.DMCI_N, z_channel=arch.DMCI_CH_Z):
        if N != arch.DMCI_N or z_channel != arch.DMCI_CH_Z:
            raise DcvcError("only the published DCVC-RT-Intra configuration (N=256, z=128) is built")
        super().__init__("dmci", z_channel, arch.QP_NUM)

    def _build_layers(self, sd, dt):
        D, C2, R, U = L.DepthConvBlock, L.Conv2d, L.ResidualBlockWithStride2, L.ResidualBlockUpsample
        n = {}
        n["enc_1"] = D(sd, "enc.enc_1", dt)
        n["enc_2"] = [D(sd, f"enc.enc_2.{i}", dt) for i in range(6)]
        n["enc_down"] = C2(sd, "enc.enc_2.6", dt, 2, 1)
        n["hyper_enc"] = [D(sd, "hyper_enc.0", dt), R(sd, "hyper_enc.1", dt), R(sd, "hyper_enc.2", dt)]
        n["hyper_dec"] = [U(sd, "hyper_dec.0", dt), U(sd, "hyper_dec.1", dt), D(sd, "hyper_dec.2", dt)]
        n["fusion"] = [D(sd, f"y_prior_fusion.{i}", dt) for i in range(3)]
        n["fusion_out"] = C2(sd, "y_prior_fusion.3", dt)
        n["reduction"] = C2(sd, "y_spatial_prior_reduction", dt)
        n["sp_adaptor"] = [None] + [D(sd, f"y_spatial_prior_adaptor_{i}", dt) for i in (1, 2, 3)]
        n["spatial"] = [D(sd, f"y_spatial_prior.{i}", dt) for i in range(3)]
        n["spatial_out"] = C2(sd, "y_spatial_prior.3", dt)
        n["dec_up"] = U(sd, "dec.dec_1.0", dt)
        n["dec_1"] = [D(sd, f"dec.dec_1.{i}", dt) for i in range(1, 13)]
        n["dec_2"] = D(sd, "dec.dec_2", dt)
        return n

    def _enc(self, x, q):
        n = self._layers
        o = n["enc_1"](self._unshuffle8(x), quant=q)
        for blk in n["enc_2"]:
            o = blk(o)
        return n["enc_down"](o)

    def _dec(self, y_hat, q):
        n = self._layers
        o = n["dec_up"](y_hat)
        for blk in n["dec_1"][:-1]:
            o = blk(o)
        o = n["dec_1"][-1](o, quant=q)
        return self._shuffle8_clamp(n["dec_2"](o))

    def _prior_params(self, z_hat, yh, yw):
        n = self._layers
        p = n["hyper_dec"][2](n["hyper_dec"][1](n["hyper_dec"][0](z_hat)))
        for blk in n["fusion"]:
            p = blk(p)
        return self._crop(n["fusion_out"](p), yh, yw)     # [yh, yw, 544]: q_enc q_dec | scales | means | pad

    def _spatial_prior(self, y_hat, common, step):
        n = self._layers
        x = n["sp_adaptor"][step](y_hat, common)
        for blk in n["spatial"]:
            x = blk(x)
        return n["spatial_out"](x)

    def compress(self, x, qp):
        """image_model.py:143-185 + compress_prior_4x (common_model.py:206-256)"""
        dtype, device = self._ensure_layers()
        n = self._layers
        C = arch.DMCI_N
        x = x.to(device=device, dtype=dtype)
        y = self._enc(x, self._q["q_scale_enc"][qp])
        yh, yw = y.shape[0], y.shape[1]
        z = n["hyper_enc"][2](n["hyper_enc"][1](n["hyper_enc"][0](self._pad_for_y(y))))
        z_hat, z8 = self._quantize_z(z)
        params = self._prior_params(z_hat, yh, yw)
        common = n["reduction"](params)
        scales, means = params[:, :, 2:2 + C], params[:, :, 2 + C:2 + 2 * C]

        nsym = (C // 4) * yh * yw
        y_hat = torch.empty((yh, yw, C), dtype=dtype, device=device)
        packed = torch.empty((4, nsym), dtype=torch.int16, device=device)
        self._prior_enc_step(4, 0, 1, y, params, scales, means, y_hat, packed[0])
        for step in (1, 2, 3):
            sp = self._spatial_prior(y_hat, common, step)
            self._prior_enc_step(4, step, 1, y, params, sp[:, :, :C], sp[:, :, C:], y_hat, packed[step])
        self._prior_finish(1, y_hat, params)

        hz = self._d2h("z8", z8)
        hp = self._d2h("packed", packed)
        ready = torch.cuda.Event()
        ready.record()
        x_hat = self._dec(y_hat, self._q["q_scale_dec"][qp])

        ready.synchronize()
        ec = self.entropy_coder
        ec.reset()
        ec.encode_z(hz.view(np.int8, z8.numel()), self._z_group, qp * self.z_channel, z.shape[0] * z.shape[1])
        ps = hp.view(np.int16, 4 * nsym)
        for k in range(4):
            ec.encode_y(ps[k * nsym:(k + 1) * nsym], self._g_group, borrowed=True)
        ec.flush()
        bit_stream = ec.get_encoded_stream()
        return {"bit_stream": bit_stream, "x_hat": x_hat}

    def decompress(self, bit_stream, sps, qp):
        """image_model.py:187-209 + decompress_prior_4x (common_model.py:258-296)"""
        dtype, device = self._ensure_layers()
        n = self._layers
        C = arch.DMCI_N
        ec = self.entropy_coder
        ec.set_use_two_entropy_coders(sps["ec_part"] == 1)
        ec.set_stream(bit_stream)
        zh, zw = self.get_downsampled_shape(sps["height"], sps["width"], 64)
        yh, yw = self.get_downsampled_shape(sps["height"], sps["width"], 16)
        nz = self.z_channel * zh * zw
        ec.decode_z(nz, self._z_group, qp * self.z_channel, zh * zw)
        zb = ec.pinned("z_dec", nz)
        ec.get_decoded(zb.view(np.int8, nz))
        z_hat = self._z_to_device(zb.view(np.int8, nz), zh, zw, dtype, device)
        params = self._prior_params(z_hat, yh, yw)
        common = n["reduction"](params)
        y_hat = torch.empty((yh, yw, C), dtype=dtype, device=device)
        self._decode_step(4, 0, params[:, :, 2:2 + C], params[:, :, 2 + C:2 + 2 * C], y_hat, yh, yw, C, "i0")
        for step in (1, 2, 3):
            sp = self._spatial_prior(y_hat, common, step)
            self._decode_step(4, step, sp[:, :, :C], sp[:, :, C:], y_hat, yh, yw, C, f"i{step}")
        self._prior_finish(1, y_hat, params)
        x_hat = self._dec(y_hat, self._q["q_scale_dec"][qp])
        return {"x_hat": x_hat}
