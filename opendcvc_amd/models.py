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
import os

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


class _CaptureGuard:
    """Process-wide reader/writer gate around HIP graph capture.

    A capture in flight is invalidated by a device synchronisation, an allocation or a free issued by ANOTHER
    host thread (the two-stage pipeline runs the encoder and the decoder on two threads).  Every per-frame
    call of a codec runs inside `frame()` (shared); a capture takes the gate exclusively: it waits until the
    other threads have left their current frame and keeps them from starting the next one until the capture
    has ended.  Work those threads already queued on their streams keeps running on the GPU - only their host
    API calls are held back.  Uncontended cost: one lock round trip per frame."""

    def __init__(self):
        import threading
        self._cv = threading.Condition()
        self._active = 0          # threads inside a frame
        self._capturing = False
        self._local = threading.local()

    def frame(self):
        return _FrameScope(self)

    def _enter(self):
        depth = getattr(self._local, "depth", 0)
        self._local.depth = depth + 1
        if depth:
            return
        with self._cv:
            while self._capturing:
                self._cv.wait()
            self._active += 1

    def _exit(self):
        self._local.depth -= 1
        if self._local.depth:
            return
        with self._cv:
            self._active -= 1
            self._cv.notify_all()

    def capture(self):
        return _CaptureScope(self)

    def _begin_capture(self):
        inside = getattr(self._local, "depth", 0) > 0
        with self._cv:
            if inside:
                self._active -= 1             # do not wait for ourselves; lets another capturer go first
                self._cv.notify_all()
            while self._capturing or self._active > 0:
                self._cv.wait()
            self._capturing = True
        return inside

    def _end_capture(self, inside):
        with self._cv:
            self._capturing = False
            if inside:
                self._active += 1
            self._cv.notify_all()


class _FrameScope:
    def __init__(self, g):
        self.g = g

    def __enter__(self):
        self.g._enter()

    def __exit__(self, *exc):
        self.g._exit()


class _CaptureScope:
    def __init__(self, g):
        self.g = g

    def __enter__(self):
        self.inside = self.g._begin_capture()

    def __exit__(self, *exc):
        self.g._end_capture(self.inside)


CAPTURE_GUARD = _CaptureGuard()


class _ModelScope:
    """One per-frame call of one model: inside CAPTURE_GUARD.frame(), and the only thread inside this model instance.
    A model's captured runs share its scratch buffers, its branch stream and its persistent staging (like the reference's
    models share their DPB): two host threads may drive two DIFFERENT instances concurrently (the two-stage pipeline does),
    never the same one - that is refused here instead of corrupting a layer's scratch between two of its kernels."""

    def __init__(self, model):
        self.m = model

    def __enter__(self):
        if not self.m._owner.acquire(blocking=False):
            raise DcvcError("this model instance is already inside compress / decompress on another host thread: one "
                            "instance is driven by one thread at a time (use one instance per thread; they may share a GPU)")
        CAPTURE_GUARD._enter()

    def __exit__(self, *exc):
        CAPTURE_GUARD._exit()
        self.m._owner.release()


# DCVC_DEC_COMPACT=0: the decoder hands whole index / symbol arrays over by copy commands (round 3's path; A/B measurements)
DEC_COMPACT = os.environ.get("DCVC_DEC_COMPACT", "1") != "0"


class _CompactStep:
    """one checkerboard decoding step's hand-off in the compacted form: pinned kept-index / count buffers, the device-side
    index array and workspace the restore needs again, later the pinned decoded symbols"""
    __slots__ = ("buf", "cnt", "idx", "ws", "cap", "sym")

    def __init__(self, buf, cnt, idx, ws, cap):
        self.buf, self.cnt, self.idx, self.ws, self.cap, self.sym = buf, cnt, idx, ws, cap, None


# DCVC_NO_FORK=1: no second stream inside a run (the temporal prior encoder then runs behind the hyper decoder)
_FORK = os.environ.get("DCVC_NO_FORK") != "1"
_PICTURE_RING = os.environ.get("DCVC_PICTURE_RING") == "1"


class GraphCache:
    """Fixed runs of kernel launches, captured once per key as a HIP graph and replayed afterwards.

    A P frame is ~130 launches in 2 (encoder) or 5 (decoder) fixed runs separated only by the host
    entropy coder; one hipGraphLaunch per run replaces ~10 us of host work per kernel and lets the
    command processor dispatch back to back (measured: 8 kernels 0.076 -> 0.017 ms host, 0.279 -> 0.262 ms
    GPU).  `fn` must be a pure function of buffers whose ADDRESSES never change between calls (graph
    outputs of earlier runs, the model's persistent staging buffers, pinned host buffers); everything it
    allocates comes from the graph's private pool.  The first call for a key runs `fn` eagerly (loads the
    code objects, sizes scratch / pinned staging), captures it and replays the capture - so `fn` must also
    be idempotent on every buffer it did not allocate itself (no in-place update of an earlier run's output).
    DCVC_NO_GRAPHS=1 keeps the plain stream launches (same kernels)."""

    def __init__(self):
        self._entries = {}
        self._side = None
        self.enabled = os.environ.get("DCVC_NO_GRAPHS") != "1"
        self._skip = set((os.environ.get("DCVC_NO_GRAPHS") or "").split(","))   # debugging: named runs stay eager

    def run(self, key, fn):
        if not self.enabled or key[0] in self._skip:
            return fn()
        e = self._entries.get(key)
        if e is None:
            fn()
            if self._side is None:      # capture needs a non-default stream; replay runs on the caller's
                self._side = torch.cuda.Stream()
            g = torch.cuda.CUDAGraph()
            with CAPTURE_GUARD.capture():       # no other host thread touches the device while this captures
                with torch.cuda.graph(g, stream=self._side, capture_error_mode="thread_local"):
                    out = fn()
            e = self._entries[key] = (g, out)
        e[0].replay()
        return e[1]

    def clear(self):
        self._entries.clear()

    def variants(self, name):
        """the variant tags (second key element) a run has been captured for"""
        return {k[1] for k in self._entries if k[0] == name}


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
        self._z_master = None
        self._layers = None
        self._layers_key = None
        self._q = {}
        self._graphs = GraphCache()
        self._persist = {}
        import threading
        self._owner = threading.RLock()           # see _ModelScope

    def _frame(self):
        return _ModelScope(self)

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
        old = self.entropy_coder
        self.entropy_coder = entropy.EntropyCoder()
        self.entropy_coder.adopt_pinned(old)      # pinned staging outlives the coder that allocated it
        self._graphs.clear()                      # the threshold is baked into the captured runs
        self._g_group = self.entropy_coder.add_cdf(*entropy.gaussian_cdf_tables())
        sd = self.state_dict()
        pre = "bit_estimator_z."
        params = {k[len(pre):]: v.detach().float().cpu() for k, v in sd.items() if k.startswith(pre)}
        if self._z_master is None and any(v.dtype != torch.float32 for k, v in sd.items() if k.startswith(pre)):
            # Only fp16 copies of bit_estimator_z exist (a half checkpoint, or another model's .half() state_dict): the
            # tables are built from what there is - exactly what the reference does with such a model - and will differ
            # from those of the fp32 checkpoint.  DCVC_STRICT_Z_TABLES=1 turns the warning into an error.
            msg = ("update(): only fp16 copies of bit_estimator_z are left (half checkpoint, or converted without half() / "
                   "load_state_dict of fp32 tensors): the z CDF tables are built from the fp16 values and may differ from "
                   "the tables of the fp32 checkpoint - streams then decode only with a decoder built the same way")
            if os.environ.get("DCVC_STRICT_Z_TABLES") == "1":
                raise DcvcError(msg)
            import warnings
            warnings.warn(msg, RuntimeWarning, stacklevel=2)
        params.update(self._z_master or {})       # fp32 master values: same tables before and after the conversion
        self._z_group = self.entropy_coder.add_cdf(*entropy.factorized_cdf_tables(params, self.qp_total, self.z_channel))

    def half(self):
        """The CDF tables of the z prior define the bit stream: they are always built from the fp32 parameters
        (the reference builds them in update() before .half(), test_video.py:398-404); a copy of those parameters
        survives the conversion, so update() may be called again afterwards and still gives the same tables."""
        pre = "bit_estimator_z."
        if self._z_master is None and next(self.parameters()).dtype == torch.float32:
            self._z_master = {k[len(pre):]: v.detach().float().cpu().clone() for k, v in self.state_dict().items()
                              if k.startswith(pre)}
        return super().half()

    def load_state_dict(self, state_dict, *args, **kwargs):
        """The incoming checkpoint's fp32 bit_estimator_z.* tensors are kept as the master copy for the CDF tables, so a
        model that is already half() - or gets converted by .to(torch.float16) / _apply instead of .half() - still
        builds the reference's tables (the reference builds them in fp32 before .half(), test_video.py:398-404).
        The master is replaced only after the load has succeeded; a dict that carries only fp16 z tensors keeps the existing
        master if those tensors are its fp16 rounding (the model's own .half() state_dict coming back), else drops it."""
        pre = "bit_estimator_z."
        z_in = {k[len(pre):]: v for k, v in state_dict.items() if k.startswith(pre) and torch.is_tensor(v)}
        res = super().load_state_dict(state_dict, *args, **kwargs)
        master = {k: v.detach().float().cpu().clone() for k, v in z_in.items() if v.dtype == torch.float32}
        if master:
            self._z_master = master
        elif z_in and self._z_master is not None:
            same = set(z_in) == set(self._z_master) and all(
                z_in[k].dtype == torch.float16 and z_in[k].shape == self._z_master[k].shape and
                torch.equal(z_in[k].detach().cpu(), self._z_master[k].half()) for k in z_in)
            if not same:
                self._z_master = None
        return res

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
            self._graphs.clear()
            self._persist = {}
            # all q_* tables side by side: one row copy per frame stages the frame's quantisation vectors at
            # fixed addresses, so the captured runs do not depend on qp
            names = sorted(self._q)
            self._q_cat = torch.cat([self._q[k] for k in names], dim=1).contiguous() if names else None
            self._q_row = torch.zeros_like(self._q_cat[0]) if names else None
            self._qv, off = {}, 0
            for k in names:
                n = self._q[k].shape[1]
                self._qv[k] = self._q_row[off:off + n]
                off += n
        return dtype, device

    def _stage_q(self, qp):
        """the frame's rows of the quantisation tables -> fixed addresses (what the captured runs read): one small kernel
        (a runtime copy command costs ~40 us of stream time, profiles/r02_launches_per_pair.txt)"""
        row = self._q_cat[qp]
        check(_lib.lib().dcvc_copy_f32(L._p(self._q_row), L._p(row), row.numel(), self._stream()), "copy_f32")
        return self._qv

    def _buffer(self, name, shape, dtype, device):
        """persistent device buffer (fixed address: graph runs read / write it across frames)"""
        key = (name, tuple(shape), dtype)
        b = self._persist.get(key)
        if b is None:
            b = self._persist[key] = torch.zeros(shape, dtype=dtype, device=device)
        return b

    def _picture_out(self, head):
        """PixelShuffle(8) + clamp of the last conv's output into a FRESH tensor, launched outside the captured run (whose own
        output buffer is overwritten by the next frame).  The reference returns a fresh tensor per frame and callers keep
        them (a sequence's pictures for the PSNR, the DPB): no ring of picture buffers - the kernel writes the new tensor
        directly, so there is no copy either, and the allocator hands the block back once the caller drops the picture.
        (DCVC_PICTURE_RING=1, a developer switch for A/B timing only: round 3's two alternating buffers - a picture is
        then overwritten by the second following frame.)"""
        if _PICTURE_RING:
            H, W, _, _ = L._geom(head)
            self._pic_parity = getattr(self, "_pic_parity", 0) ^ 1
            buf = self._buffer(f"picture_{self._pic_parity}", (1, 3, H * 8, W * 8), head.dtype, head.device)
            return self._shuffle8_clamp(head, out=buf)
        return self._shuffle8_clamp(head)

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

    def _unshuffle8(self, x, out=None):
        _, C, H, W = x.shape
        x = x.contiguous()
        if out is None:
            out = torch.empty((H // 8, W // 8, C * 64), dtype=x.dtype, device=x.device)
        check(_lib.lib().dcvc_unshuffle8(L.dtype_code(x.dtype), L._p(x), C, H, W, L._p(out), C * 64, self._stream()),
              "unshuffle8")
        return out

    def _shuffle8_clamp(self, x, bias=None, out=None):
        H, W, C, ld = L._geom(x)
        if out is None:
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

    def _z_to_device(self, z_pinned, zh, zw, dtype, device):
        """decoded z symbols (int8, CHW, in a PINNED host buffer) -> z_hat on the device: the kernel reads the pinned
        buffer in place (coalesced byte reads over the host link; no copy command)"""
        out = torch.empty((zh, zw, self.z_channel), dtype=dtype, device=device)
        check(_lib.lib().dcvc_z_from_int8(L.dtype_code(dtype), ctypes.c_void_p(z_pinned.dptr), zh, zw, self.z_channel,
                                          L._p(out), self.z_channel, self._stream()), "z_from_int8")
        return out

    # ---- host <-> device staging
    def _symbols_to_host(self, key, z8, packed):
        """Encoder hand-off without a copy command: z (int8) and the KEPT y symbols of each part of `packed`, compacted in
        order on the device, are written by kernels straight into pinned host buffers (dcvc_compact_symbols).  Returns
        (z buffer, symbol buffer, counts buffer); valid for the host once the stream has passed this point."""
        lib = _lib.lib()
        ec = self.entropy_coder
        parts, nsym = packed.shape
        nz = z8.numel()
        hz = ec.pinned(key + "_z", (nz + 3) // 4 * 4)
        hp = ec.pinned(key + "_sym", parts * nsym * 2)
        hc = ec.pinned(key + "_cnt", 4 * parts)
        ws = self._buffer("compact_ws", (256 * parts,), torch.int32, packed.device)
        if nz % 4 == 0:
            check(lib.dcvc_copy_f32(ctypes.c_void_p(hz.dptr), L._p(z8), nz // 4, self._stream()), "z to host")
        else:
            check(lib.dcvc_memcpy_d2h(ctypes.c_void_p(hz.ptr), L._p(z8), nz, self._stream()), "d2h")
        check(lib.dcvc_compact_symbols(L._p(packed), nsym, parts, ctypes.c_void_p(hp.ptr), ctypes.c_void_p(hc.ptr), L._p(ws),
                                       self._stream()), "compact_symbols")
        return hz, hp, hc

    def _prior_enc_step(self, groups, step, q_mode, y, qsrc, scales, means, yhat, packed):
        H, W, C, ldy = L._geom(y)
        check(_lib.lib().dcvc_prior_enc_step(
            L.dtype_code(y.dtype), groups, step, q_mode, L._p(y), ldy, L._p(qsrc), qsrc.stride(1), L._p(scales),
            scales.stride(1), L._p(means), means.stride(1), H, W, C, self._thres(), L._p(yhat), yhat.stride(1),
            L._p(yhat), yhat.stride(1), L._p(packed), self._stream()), "prior_enc_step")

    def _prior_finish(self, q_mode, yhat, qsrc):
        H, W, C, ld = L._geom(yhat)
        check(_lib.lib().dcvc_prior_finish(L.dtype_code(yhat.dtype), q_mode, L._p(yhat), ld, L._p(qsrc), qsrc.stride(1),
                                           H, W, C, self._stream()), "prior_finish")

    # one checkerboard decoding step = three pieces, so that the device pieces can sit inside captured runs
    # (device index build -> host rANS decode -> device restore)
    def _index_to_host(self, groups, step, scales, H, W, C, key):
        """device: cdf indexes of the step's symbols -> host.  Default (DEC_COMPACT): the KEPT indexes only, compacted in stream
        order by a kernel that writes them (and their count) straight into pinned buffers - returns a _CompactStep; the
        decoded symbols come back the same way (_symbols_to_device).  DCVC_DEC_COMPACT=0: the whole array, see below.

        Whole-array form: pinned host buffer by a stream-ordered copy.
        Measured in round 4 (profiles/r04_dec_inplace.txt): the kernel writing the indexes straight into the pinned buffer takes
        39 us instead of 5.4 us + a ~10 us copy command, and the restore kernel reading the symbols in place 108 us instead
        of 7.7 us + copy - a channel's run of 16 pixels is 16 bytes, one bus transaction per lane, where the copy moves
        whole lines; only z (read coalesced, 65 KB) is taken in place."""
        n = (C // groups) * H * W
        idx = torch.empty(n, dtype=torch.uint8, device=scales.device)
        if DEC_COMPACT:
            lib = _lib.lib()
            cap = (n + 15) // 16 * 16
            ws = torch.empty(int(lib.dcvc_prior_dec_compact_ws_bytes(H, W, C, groups)), dtype=torch.uint8, device=scales.device)
            buf = self.entropy_coder.pinned(key + "_cidx", cap)
            cnt = self.entropy_coder.pinned(key + "_ccnt", 16)
            check(lib.dcvc_prior_dec_index_compact(L.dtype_code(scales.dtype), groups, step, L._p(scales), scales.stride(1),
                                                   H, W, C, self._thres(), L._p(idx), L._p(ws), ctypes.c_void_p(buf.ptr),
                                                   ctypes.c_void_p(cnt.ptr), self._stream()), "prior_dec_index_compact")
            return _CompactStep(buf, cnt, idx, ws, cap)
        check(_lib.lib().dcvc_prior_dec_index(L.dtype_code(scales.dtype), groups, step, L._p(scales), scales.stride(1),
                                              H, W, C, self._thres(), L._p(idx), self._stream()), "prior_dec_index")
        buf = self.entropy_coder.pinned(key + "_idx", n)
        check(_lib.lib().dcvc_memcpy_d2h(ctypes.c_void_p(buf.ptr), L._p(idx), n, self._stream()), "d2h")
        return buf

    def _decode_on_host(self, idx_host, n, key):
        """host: rANS-decode the step's symbols (the caller has waited for the stream to pass the index hand-off)"""
        if isinstance(idx_host, _CompactStep):
            cs = idx_host
            count = int(cs.cnt.view(np.int32, 1)[0])
            if not 0 <= count <= n:
                raise DcvcError("decoder hand-off: %d kept symbols of %d positions" % (count, n))
            cs.sym = self.entropy_coder.pinned(key + "_csym", cs.cap)
            self.entropy_coder.decode_compact(cs.buf.view(np.uint8, cs.cap), count, self._g_group, cs.sym.view(np.int8, cs.cap))
            return cs
        sb = self.entropy_coder.pinned(key + "_sym", n)
        self.entropy_coder.decode_and_get_y(idx_host.view(np.uint8, n), self._g_group, sb.view(np.int8, n))
        return sb

    def _symbols_to_device(self, sym_host, n, groups, step, means, yhat, H, W, C, out=None):
        """device: upload the decoded symbols (stream-ordered copy, see _index_to_host) and restore y_hat at the step's
        positions"""
        out = yhat if out is None else out
        if isinstance(sym_host, _CompactStep):
            cs = sym_host
            check(_lib.lib().dcvc_prior_dec_restore_compact(
                L.dtype_code(means.dtype), groups, step, ctypes.c_void_p(cs.sym.ptr), L._p(cs.idx), L._p(cs.ws), L._p(means),
                means.stride(1), H, W, C, L._p(yhat), yhat.stride(1), L._p(out), out.stride(1), self._stream()),
                "prior_dec_restore_compact")
            return
        sym = torch.empty(n, dtype=torch.int8, device=yhat.device)
        check(_lib.lib().dcvc_memcpy_h2d(L._p(sym), ctypes.c_void_p(sym_host.ptr), n, self._stream()), "h2d")
        check(_lib.lib().dcvc_prior_dec_restore(L.dtype_code(means.dtype), groups, step, L._p(sym), L._p(means),
                                                means.stride(1), H, W, C, L._p(yhat), yhat.stride(1), L._p(out),
                                                out.stride(1), self._stream()), "prior_dec_restore")


# =============================================================================== DMC (P frames)

class DMC(CompressionModel):
    """reference: DMC (video_model.py:226-379)"""

    def __init__(self):
        super().__init__("dmc", arch.DMC_CH_Z, arch.QP_NUM + arch.DMC_EXTRA_QP)
        self.qp_shift = list(arch.DMC_QP_SHIFT)
        self.dpb = []
        self.max_dpb_size = 1
        self.curr_poc = 0
        self._ahead = None       # (height, width, x1, ctx): feature extractor of the NEXT frame, run ahead
        self._pending = None     # decoder, deferred output: the reconstruction network of the previous frame still to run
        self._stream_pending = None   # encoder, deferred stream: symbols of the previous frame still to be entropy coded
        self._stream_parity = 0
        self._branch = None      # second HIP stream for the branches inside a run (_prior_params)

    def _build_layers(self, sd, dt):
        D, C2, R = L.DepthConvBlock, L.Conv2d, L.ResidualBlockWithStride2
        n = {}
        n["fa_i"] = D(sd, "feature_adaptor_i", dt)
        # feature_adaptor_p (1x1, 256 -> 256) followed by the extractor's first DepthConvBlock IS a block with an
        # adaptor: one head launch computes f = adaptor_p(ref) (written out as the block's identity branch) and the
        # first conv + activation on the tile still in LDS - same kernels' arithmetic, same roundings as the two launches
        fused = {"m" + k[len("feature_extractor.conv1.0"):]: v for k, v in sd.items() if k.startswith("feature_extractor.conv1.0.")}
        fused["m.adaptor.weight"], fused["m.adaptor.bias"] = sd["feature_adaptor_p.weight"], sd["feature_adaptor_p.bias"]
        n["fe1_p"] = [D(fused, "m", dt)]
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
        # y_spatial_prior's last conv is 384 -> 256: as a SQUARE 384 -> 384 conv (128 all-zero output rows, never read) it qualifies
        # for the fused launch inside the preceding block's tail (Conv2d.fusable_after) - the same dot products for the real
        # channels, one 12 - 14 us launch less per coded frame on the decoder's critical path (DCVC_SQUARE_SPATIAL_OUT=0: as is)
        w, b = sd["y_spatial_prior.conv.2.weight"], sd["y_spatial_prior.conv.2.bias"]
        if os.environ.get("DCVC_SQUARE_SPATIAL_OUT") != "0" and w.shape[0] < w.shape[1] and w.shape[2:] == (1, 1):
            wp = torch.zeros((w.shape[1], w.shape[1], 1, 1), dtype=w.dtype)
            wp[:w.shape[0]] = w
            bp = torch.zeros(w.shape[1], dtype=b.dtype)
            bp[:b.shape[0]] = b
            n["spatial_out"] = C2({"c.weight": wp, "c.bias": bp}, "c", dt)
        else:
            n["spatial_out"] = C2(sd, "y_spatial_prior.conv.2", dt)
        n["dec_up"] = L.SubpelConv2x(sd, "decoder.up", dt, 1)
        n["dec_conv1"] = [D(sd, f"decoder.conv1.{i}", dt) for i in range(3)]
        n["dec_conv2"] = C2(sd, "decoder.conv2", dt, epilogue=_lib.EPI_BIAS_QUANT)
        n["recon"] = [D(sd, f"recon_generation_net.conv.{i}", dt) for i in range(4)]
        n["recon_head"] = C2(sd, "recon_generation_net.head", dt)
        return n

    # ---- DPB (video_model.py:253-277)
    def reset_ref_feature(self):
        self._ahead = None
        if len(self.dpb) > 0:
            self.dpb[0].feature = None

    def add_ref_frame(self, feature=None, frame=None, increase_poc=True):
        self._ahead = None           # a context computed ahead belongs to the previous reference
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
        self._ahead = None
        self.dpb.clear()     # (a deferred reconstruction stays pending: it only needs the feature buffer)

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
    def _fe1(self, variant):
        """conv1 of the feature extractor as seen from the reference buffer: after a P frame its first block carries
        feature_adaptor_p as its adaptor (fe1_p); after an I frame / a refresh feature_adaptor_i is a block of its own"""
        n = self._layers
        return n["fe1_p"] + n["fe1"][1:] if variant == "p" else [n["fa_i"]] + n["fe1"]

    def _fe_input(self, variant, ref_buf):
        return ref_buf if variant == "p" else self._unshuffle8(ref_buf)

    def _extractor_part1(self, variant, ref_buf):
        """conv1 of the feature extractor.  The reference's second output ctx_t = x1 * q_feature (video_model.py:43-47) is
        read by temporal_prior_encoder only: its first conv scales x1 while staging it (_prior_params), the tensor is
        never materialised."""
        return L.dcb_chain(self._fe1(variant), self._fe_input(variant, ref_buf))

    def _extractor_part2(self, x1):
        return L.dcb_chain(self._layers["fe2"], x1)

    def _extractor_both(self, variant, ref_buf):
        """feature adaptor, conv1 and conv2 of the feature extractor as ONE chain (encoder side, where they sit in the
        same captured run): every block starts in its predecessor's tail.  Returns (x1, ctx); same values, bit for bit,
        as part1 + part2 on the decoder (test_chained_blocks_equal_separate_calls and the enc/dec context tests)."""
        blocks = self._fe1(variant)
        outs = L.dcb_chain(blocks + self._layers["fe2"], self._fe_input(variant, ref_buf), return_all=True)
        return outs[len(blocks) - 1], outs[-1]

    def _branch_stream(self, device):
        """second stream of this model for branches inside a run.  One per model instance, like the scratch its kernels
        use (nn.Scratch is per stream): safe because one instance is inside one frame call at a time (_ModelScope) - the
        encoder's and the decoder's instances, which do run concurrently, each have their own."""
        if self._branch is None:
            self._branch = torch.cuda.Stream(device)
        return self._branch

    def _prior_params(self, z_hat, x1, q_feature, yh, yw):
        """res_prior_param_decoder (video_model.py:279-286) -> [yh, yw, 384] = q_dec | scales | means.
        x1, q_feature: temporal_prior_encoder reads ctx_t = x1 * q_feature (same values as the stand-alone product)"""
        n = self._layers
        cat = torch.empty((yh, yw, 3 * arch.DMC_CH_Y), dtype=z_hat.dtype, device=z_hat.device)
        # The hyper decoder works on 17x30 / 34x60 maps (kernels of 16 - 64 workgroups that leave the GPU empty) and the
        # temporal prior encoder does not depend on it: the two run side by side, the temporal branch on a second stream
        # (a fork / join of the captured run; its own scratch: nn.Scratch is per stream).  Same kernels, same values.
        main = torch.cuda.current_stream()
        side = self._branch_stream(z_hat.device) if _FORK else None
        if side is not None:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                n["temporal"](x1, in_scale=q_feature, out=cat[:, :, arch.DMC_CH_Y:])
        h = n["hyper_dec"][1](n["hyper_dec"][0](z_hat))
        if h.shape[0] == yh and h.shape[1] == yw:
            n["hyper_dec"][2](h, out=cat[:, :, :arch.DMC_CH_Y])
        else:
            self._crop(n["hyper_dec"][2](h), yh, yw, out=cat[:, :, :arch.DMC_CH_Y])
        if side is not None:
            main.wait_stream(side)
        else:
            n["temporal"](x1, in_scale=q_feature, out=cat[:, :, arch.DMC_CH_Y:])
        return L.dcb_chain(n["fusion"], cat, then_conv=n["fusion_out"])          # (the last conv runs in the last tail)

    def _spatial_prior(self, y_hat, params):
        n = self._layers
        return L.dcb_chain(n["spatial"], y_hat, params, then_conv=n["spatial_out"])

    def _decoder(self, y_hat, ctx, q_decoder, out=None):
        n = self._layers
        return L.dcb_chain(n["dec_conv1"], n["dec_up"](y_hat), ctx, then_conv=n["dec_conv2"], conv_quant=q_decoder, out=out)

    def _recon_head(self, feature, q_recon):
        """recon_generation_net up to its last conv (video_model.py:151-163); _picture_out() makes the picture of it"""
        n = self._layers
        o = L.dcb_chain(n["recon"], feature, quant=q_recon)
        return n["recon_head"](o)

    def _recon(self, feature, q_recon):
        return self._picture_out(self._recon_head(feature, q_recon))

    # the same network in two halves (deferred decoder output: each half fills one host-decoding gap of the next frame)
    def _recon_first(self, feature):
        return L.dcb_chain(self._layers["recon"][:2], feature)

    def _recon_second(self, mid, q_recon):
        n = self._layers
        return n["recon_head"](L.dcb_chain(n["recon"][2:], mid, quant=q_recon))      # (the picture: _picture_out)

    def _pending_half(self, which):
        """launches one half of the pending reconstruction (no-op when nothing is pending / already done)"""
        pd = self._pending
        if pd is None:
            return
        if which == 0 and pd["mid"] is None:
            pd["mid"] = self._graphs.run(("dec_ra",) + pd["key"], lambda: self._recon_first(pd["feature"]))
        elif which == 1 and pd["mid"] is not None and pd["x_hat"] is None:
            mid, qrec = pd["mid"], pd["q_recon"]
            head = self._graphs.run(("dec_rb",) + pd["key"], lambda: self._recon_second(mid, qrec))
            pd["x_hat"] = self._picture_out(head)

    def finish_output(self):
        """finish_output inside this model's frame scope (_ModelScope): while another thread's GraphCache.run is capturing, a frame
        neither synchronises nor allocates (so two model INSTANCES may be driven from two host threads, through this
        reference-compatible API as well as through SequenceEncoder / SequenceDecoder); a second thread entering the SAME
        instance is refused.  The scope is re-entrant per thread."""
        with self._frame():
            return self._finish_output_unguarded()

    def _finish_output_unguarded(self):
        """Deferred decoder output: completes and returns the reconstruction of the last decompress(...,
        defer_output=True) (None if there is none).  Called implicitly by the next decompress."""
        if self._pending is None:
            return None
        self._pending_half(0)
        self._pending_half(1)
        x_hat = self._pending["x_hat"]
        self._pending = None
        if self.dpb and self.dpb[0].frame is None:
            self.dpb[0].frame = x_hat
        return x_hat

    # ---- frame API
    def _stage_reference(self, dtype, device):
        """Puts the reference (feature, or the reconstructed frame after an I frame / a refresh) where the
        captured runs expect it; returns the variant key."""
        ref = self.dpb[0]
        if ref.feature is None:
            f = ref.frame
            buf = self._buffer("ref_frame", f.shape, dtype, device)
            buf.copy_(f, non_blocking=True)
            return "i", buf
        if ref.feature.data_ptr() != self._feature_buf(ref.feature.shape, dtype, device).data_ptr():
            self._feature_buf(ref.feature.shape, dtype, device).copy_(ref.feature, non_blocking=True)
        return "p", self._feature_buf(ref.feature.shape, dtype, device)

    def _feature_buf(self, shape, dtype, device):
        return self._buffer("feature", shape, dtype, device)

    def _code_symbols(self, job):
        """host: entropy-codes one frame's symbols (waits for their copy to the pinned staging first)"""
        ready, hz, hp, hc, nz, nsym, zhw, qp = job
        ready.synchronize()
        ec = self.entropy_coder
        ec.reset()
        ec.encode_z(hz.view(np.int8, nz), self._z_group, qp * self.z_channel, zhw)
        ps, kept = hp.view(np.int16, 2 * nsym), hc.view(np.int32, 2)
        ec.encode_y(ps[:kept[0]], self._g_group, borrowed=True)                 # pinned staging buffer, untouched until
        ec.encode_y(ps[nsym:nsym + kept[1]], self._g_group, borrowed=True)     # get_encoded_stream() below
        ec.flush()
        return ec.get_encoded_stream()

    def finish_stream(self):
        """finish_stream inside this model's frame scope (_ModelScope): while another thread's GraphCache.run is capturing, a frame
        neither synchronises nor allocates (so two model INSTANCES may be driven from two host threads, through this
        reference-compatible API as well as through SequenceEncoder / SequenceDecoder); a second thread entering the SAME
        instance is refused.  The scope is re-entrant per thread."""
        with self._frame():
            return self._finish_stream_unguarded()

    def _finish_stream_unguarded(self):
        """Deferred encoder stream: entropy-codes and returns the bit stream of the last compress(..., defer_stream=True)
        (None if there is none).  Called implicitly by the next compress."""
        job, self._stream_pending = self._stream_pending, None
        return None if job is None else self._code_symbols(job)

    def compress(self, x, qp, defer_stream=False):
        """compress inside this model's frame scope (_ModelScope): while another thread's GraphCache.run is capturing, a frame
        neither synchronises nor allocates (so two model INSTANCES may be driven from two host threads, through this
        reference-compatible API as well as through SequenceEncoder / SequenceDecoder); a second thread entering the SAME
        instance is refused.  The scope is re-entrant per thread."""
        with self._frame():
            return self._compress_unguarded(x, qp, defer_stream=defer_stream)

    def _compress_unguarded(self, x, qp, defer_stream=False):
        """video_model.py:299-341.  x: [1,3,H,W] in [0,1], H and W multiples of 16.
        Two captured runs: everything up to the symbol hand-off, then the decoder (which overlaps the host
        entropy coding).

        defer_stream=True (not in the reference API; the encoder-side mirror of decompress(defer_output=True)): the
        symbols of THIS frame are entropy coded during the next call, underneath that frame's kernels - the returned
        dict then carries the PREVIOUS frame's stream under 'bit_stream_prev' and 'bit_stream' is None;
        finish_stream() returns the last one.  Same symbols, same bytes; a sequential encoder becomes GPU-bound."""
        dtype, device = self._ensure_layers()
        n = self._layers
        C = arch.DMC_CH_Y
        x = x.to(device=device, dtype=dtype)
        _, _, H, W = x.shape
        xin = self._buffer("x_unshuffled", (H // 8, W // 8, 192), dtype, device)
        self._unshuffle8(x, out=xin)
        q = self._stage_q(qp)
        variant, ref_buf = self._stage_reference(dtype, device)
        fbuf = self._feature_buf((H // 8, W // 8, arch.DMC_CH_D), dtype, device)
        key = (variant, H, W)
        # The feature extractor depends on the reference feature only (not on this frame, not on qp): the previous
        # compress() has already run it for us while its own entropy coding kept the host busy.
        ahead = self._ahead if (variant == "p" and self._ahead is not None and self._ahead[:2] == (H, W)) else None

        def front():
            if ahead is not None:
                x1, ctx = ahead[2], ahead[3]
            else:
                x1, ctx = self._extractor_both(variant, ref_buf)
            e = L.dcb_chain(n["enc_conv2"] + [n["enc_conv3"]], n["enc_conv1"](xin), ctx, quant=q["q_encoder"])
            y = n["enc_down"](e)
            yh, yw = y.shape[0], y.shape[1]
            z = n["hyper_enc"][2](n["hyper_enc"][1](n["hyper_enc"][0](self._pad_for_y(y))))
            z_hat, z8 = self._quantize_z(z)
            params = self._prior_params(z_hat, x1, q["q_feature"], yh, yw)
            nsym = (C // 2) * yh * yw
            y_hat = torch.empty((yh, yw, C), dtype=dtype, device=device)
            packed = torch.empty((2, nsym), dtype=torch.int16, device=device)
            self._prior_enc_step(2, 0, 0, y, params[:, :, :C], params[:, :, C:2 * C], params[:, :, 2 * C:], y_hat, packed[0])
            sp = self._spatial_prior(y_hat, params)
            self._prior_enc_step(2, 1, 0, y, params[:, :, :C], sp[:, :, :C], sp[:, :, C:], y_hat, packed[1])
            self._prior_finish(0, y_hat, params[:, :, :C])
            return y_hat, ctx, z8, packed, z8.numel(), nsym, (z.shape[0], z.shape[1])

        y_hat, ctx, z8, packed, nz, nsym, (zh, zw) = self._graphs.run(
            ("enc_front_ahead" if ahead is not None else "enc_front",) + key, front)
        # symbols -> pinned staging, compacted, by kernels (outside the captured run: two staging sets alternate, so that
        # the host may still be coding the previous frame out of the other one while these writes land)
        par = self._stream_parity
        self._stream_parity ^= 1
        hz, hp, hc = self._symbols_to_host(f"enc{par}", z8, packed)
        ready = torch.cuda.Event()
        ready.record()
        # the decoder keeps the GPU busy while the host codes
        # (keyed by the front run too: its y_hat / ctx are that run's static outputs)
        self._graphs.run(("enc_back", "ahead" if ahead is not None else "full") + key,
                         lambda: self._decoder(y_hat, ctx, q["q_decoder"], out=fbuf))
        nxt = None
        if self._graphs.enabled:     # next frame's extractor, behind the decoder and under the host coder below
            def extractor_ahead():
                return self._extractor_both("p", fbuf)
            nxt = (H, W) + tuple(self._graphs.run(("enc_ahead", H, W), extractor_ahead))

        # host entropy coding: the previous frame's deferred symbols first (its staging set is reused two frames on)
        prev = self.finish_stream()
        job = (ready, hz, hp, hc, nz, nsym, zh * zw, qp)
        # no device synchronisation here (the reference has none either): the tail of the decoder stays in
        # flight on this stream and overlaps the caller's next host work; callers that time a frame sync.
        self.add_ref_frame(fbuf, None)
        self._ahead = nxt
        if defer_stream:
            self._stream_pending = job
            return {"bit_stream": None, "bit_stream_prev": prev}
        bit_stream = self._code_symbols(job)
        return {"bit_stream": bit_stream} if prev is None else {"bit_stream": bit_stream, "bit_stream_prev": prev}

    def decompress(self, bit_stream, sps, qp, defer_output=False):
        """decompress inside this model's frame scope (_ModelScope): while another thread's GraphCache.run is capturing, a frame
        neither synchronises nor allocates (so two model INSTANCES may be driven from two host threads, through this
        reference-compatible API as well as through SequenceEncoder / SequenceDecoder); a second thread entering the SAME
        instance is refused.  The scope is re-entrant per thread."""
        with self._frame():
            return self._decompress_unguarded(bit_stream, sps, qp, defer_output=defer_output)

    def _decompress_unguarded(self, bit_stream, sps, qp, defer_output=False):
        """video_model.py:343-376.  Five captured runs, separated by the three host decoding steps
        (z, first and second checkerboard half).

        A damaged payload raises DcvcError after the frame's last symbol (entropy coder end-state check); the frame is then
        not added to the DPB (a pending deferred picture of the PREVIOUS frame stays retrievable through finish_output()):
        resume at the next I frame (clear_dpb()).

        defer_output=True (not in the reference API): the reconstruction network of THIS frame is not run now
        but in the two host-decoding gaps of the next call (nothing else can use the GPU there: the next
        frame's symbols are not known yet) - the returned dict then carries the PREVIOUS frame under
        'x_hat_prev' and 'x_hat' is None; finish_output() returns the last frame.  Same kernels, same values."""
        dtype, device = self._ensure_layers()
        C = arch.DMC_CH_Y
        forced = None
        if self._pending is not None and (not defer_output or (self.dpb and self.dpb[0].feature is None)):
            forced = self.finish_output()          # the refresh path needs the previous picture itself
        prev = self._pending
        ec = self.entropy_coder
        ec.set_use_two_entropy_coders(sps["ec_part"] == 1)
        ec.set_stream(bit_stream)
        zh, zw = self.get_downsampled_shape(sps["height"], sps["width"], 64)
        yh, yw = self.get_downsampled_shape(sps["height"], sps["width"], 16)
        nz = self.z_channel * zh * zw
        ec.decode_z(nz, self._z_group, qp * self.z_channel, zh * zw)     # host worker, overlaps the first run

        q = self._stage_q(qp)
        variant, ref_buf = self._stage_reference(dtype, device)
        fbuf = self._feature_buf((2 * yh, 2 * yw, arch.DMC_CH_D), dtype, device)
        key = (variant, sps["height"], sps["width"])
        n_half = (C // 2) * yh * yw
        zb = ec.pinned("z_dec", nz)

        x1 = self._graphs.run(("dec_0",) + key, lambda: self._extractor_part1(variant, ref_buf))
        ec.get_decoded(zb.view(np.int8, nz))

        def after_z():
            z_hat = self._z_to_device(zb, zh, zw, dtype, device)
            params = self._prior_params(z_hat, x1, q["q_feature"], yh, yw)
            y_hat = torch.empty((yh, yw, C), dtype=dtype, device=device)
            return params, y_hat, self._index_to_host(2, 0, params[:, :, C:2 * C], yh, yw, C, "p0")

        params, y_hat, idx0 = self._graphs.run(("dec_1",) + key, after_z)
        ev = torch.cuda.Event()
        ev.record()
        ctx = self._graphs.run(("dec_2",) + key, lambda: self._extractor_part2(x1))   # overlaps the host decode
        self._pending_half(0)
        ev.synchronize()
        sym0 = self._decode_on_host(idx0, n_half, "p0")

        def after_step0():
            self._symbols_to_device(sym0, n_half, 2, 0, params[:, :, 2 * C:], y_hat, yh, yw, C)
            sp = self._spatial_prior(y_hat, params)
            return sp, self._index_to_host(2, 1, sp[:, :, :C], yh, yw, C, "p1")

        sp, idx1 = self._graphs.run(("dec_3",) + key, after_step0)
        ev = torch.cuda.Event()
        ev.record()
        self._pending_half(1)
        ev.synchronize()
        sym1 = self._decode_on_host(idx1, n_half, "p1")
        ec.check_end()                    # corrupt / truncated payload: DcvcError here, not a garbage picture

        def after_step1():
            # runs must be idempotent on buffers they did not allocate (see GraphCache): the second half is
            # accumulated into a fresh tensor, not into the first run's y_hat
            y_fin = torch.empty_like(y_hat)
            self._symbols_to_device(sym1, n_half, 2, 1, sp[:, :, C:], y_hat, yh, yw, C, out=y_fin)
            self._prior_finish(0, y_fin, params[:, :, :C])
            feature = self._decoder(y_fin, ctx, q["q_decoder"], out=fbuf)
            return None if defer_output else self._recon_head(feature, q["q_recon"])

        x_prev = None
        if prev is not None:              # both halves of the previous frame's reconstruction are in flight by now
            x_prev = prev["x_hat"]
            self._pending = None
        head = self._graphs.run(("dec_4d" if defer_output else "dec_4",) + key, after_step1)
        x_hat = None if head is None else self._picture_out(head)
        self.add_ref_frame(fbuf, x_hat)
        if defer_output:
            qrec = self._buffer("q_recon_pending", q["q_recon"].shape, torch.float32, device)
            qrec.copy_(q["q_recon"], non_blocking=True)
            self._pending = dict(key=(sps["height"], sps["width"]), feature=fbuf, q_recon=qrec, mid=None, x_hat=None)
            return {"x_hat": None, "x_hat_prev": x_prev if x_prev is not None else forced}
        return {"x_hat": x_hat} if forced is None else {"x_hat": x_hat, "x_hat_prev": forced}


DMC.encode_one_frame = DMC.compress          # the names BASELINE.json's north_star uses for the per-frame API (the reference
DMC.decode_one_frame = DMC.decompress        # itself has only compress / decompress: SURVEY.md section 0-1)


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
        return n["enc_down"](L.dcb_chain(n["enc_2"], o))

    def _dec(self, y_hat, q):
        n = self._layers
        o = L.dcb_chain(n["dec_1"], n["dec_up"](y_hat), quant=q)
        return n["dec_2"](o)             # (the picture: _picture_out)

    def _prior_params(self, z_hat, yh, yw):
        n = self._layers
        p = n["hyper_dec"][2](n["hyper_dec"][1](n["hyper_dec"][0](z_hat)))
        return self._crop(L.dcb_chain(n["fusion"], p, then_conv=n["fusion_out"]), yh, yw)     # [yh, yw, 544]: q_enc q_dec | scales | means | pad

    def _spatial_prior(self, y_hat, common, step):
        n = self._layers
        return L.dcb_chain([n["sp_adaptor"][step]] + n["spatial"], y_hat, common, then_conv=n["spatial_out"])

    def compress(self, x, qp):
        """compress inside this model's frame scope (_ModelScope): while another thread's GraphCache.run is capturing, a frame
        neither synchronises nor allocates (so two model INSTANCES may be driven from two host threads, through this
        reference-compatible API as well as through SequenceEncoder / SequenceDecoder); a second thread entering the SAME
        instance is refused.  The scope is re-entrant per thread."""
        with self._frame():
            return self._compress_unguarded(x, qp)

    def _compress_unguarded(self, x, qp):
        """image_model.py:143-185 + compress_prior_4x (common_model.py:206-256).  Two captured runs, like DMC:
        everything up to the symbol hand-off, then the synthesis transform (which overlaps the host entropy coder)."""
        dtype, device = self._ensure_layers()
        n = self._layers
        C = arch.DMCI_N
        x = x.to(device=device, dtype=dtype)
        _, _, H, W = x.shape
        xin = self._buffer("x_unshuffled", (H // 8, W // 8, 192), dtype, device)
        self._unshuffle8(x, out=xin)
        q = self._stage_q(qp)
        key = (H, W)

        def front():
            y = n["enc_down"](L.dcb_chain(n["enc_2"], n["enc_1"](xin, quant=q["q_scale_enc"])))
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
            return y_hat, z8, packed, z8.numel(), nsym, (z.shape[0], z.shape[1])

        y_hat, z8, packed, nz, nsym, (zh, zw) = self._graphs.run(("ienc_front",) + key, front)
        hz, hp, hc = self._symbols_to_host("ienc", z8, packed)
        ready = torch.cuda.Event()
        ready.record()
        x_hat = self._picture_out(self._graphs.run(("ienc_back",) + key, lambda: self._dec(y_hat, q["q_scale_dec"])))

        ready.synchronize()
        ec = self.entropy_coder
        ec.reset()
        ec.encode_z(hz.view(np.int8, nz), self._z_group, qp * self.z_channel, zh * zw)
        ps, kept = hp.view(np.int16, 4 * nsym), hc.view(np.int32, 4)
        for k in range(4):
            ec.encode_y(ps[k * nsym:k * nsym + kept[k]], self._g_group, borrowed=True)
        ec.flush()
        bit_stream = ec.get_encoded_stream()
        return {"bit_stream": bit_stream, "x_hat": x_hat}

    def decompress(self, bit_stream, sps, qp):
        """decompress inside this model's frame scope (_ModelScope): while another thread's GraphCache.run is capturing, a frame
        neither synchronises nor allocates (so two model INSTANCES may be driven from two host threads, through this
        reference-compatible API as well as through SequenceEncoder / SequenceDecoder); a second thread entering the SAME
        instance is refused.  The scope is re-entrant per thread."""
        with self._frame():
            return self._decompress_unguarded(bit_stream, sps, qp)

    def _decompress_unguarded(self, bit_stream, sps, qp):
        """image_model.py:187-209 + decompress_prior_4x (common_model.py:258-296).  Five captured runs split at the
        four host decoding steps (the reference's dependency structure: each checkerboard step needs the symbols of
        the previous one); a run never updates an earlier run's output in place (GraphCache), so every step restores
        into a fresh y_hat."""
        dtype, device = self._ensure_layers()
        n = self._layers
        C = arch.DMCI_N
        ec = self.entropy_coder
        ec.set_use_two_entropy_coders(sps["ec_part"] == 1)
        ec.set_stream(bit_stream)
        zh, zw = self.get_downsampled_shape(sps["height"], sps["width"], 64)
        yh, yw = self.get_downsampled_shape(sps["height"], sps["width"], 16)
        nz = self.z_channel * zh * zw
        nsym = (C // 4) * yh * yw
        key = (sps["height"], sps["width"])
        q = self._stage_q(qp)
        ec.decode_z(nz, self._z_group, qp * self.z_channel, zh * zw)
        zb = ec.pinned("z_dec", nz)
        ec.get_decoded(zb.view(np.int8, nz))

        def first():
            z_hat = self._z_to_device(zb, zh, zw, dtype, device)
            params = self._prior_params(z_hat, yh, yw)
            common = n["reduction"](params)
            return params, common, self._index_to_host(4, 0, params[:, :, 2:2 + C], yh, yw, C, "i0")

        params, common, idx = self._graphs.run(("idec_0",) + key, first)
        means = params[:, :, 2 + C:2 + 2 * C]
        y_prev = None
        for step in (0, 1, 2, 3):
            ev = torch.cuda.Event()
            ev.record()
            ev.synchronize()                       # the index copy of this step has landed
            sym = self._decode_on_host(idx, nsym, f"i{step}")
            if step == 3:
                ec.check_end()            # corrupt / truncated payload: DcvcError here, not a garbage picture

            def after(step=step, sym=sym, means=means, y_prev=y_prev):
                y_new = torch.empty((yh, yw, C), dtype=dtype, device=device)
                self._symbols_to_device(sym, nsym, 4, step, means, y_new if y_prev is None else y_prev, yh, yw, C, out=y_new)
                if step == 3:
                    self._prior_finish(1, y_new, params)
                    return y_new, self._dec(y_new, q["q_scale_dec"])
                sp = self._spatial_prior(y_new, common, step + 1)
                return y_new, sp, self._index_to_host(4, step + 1, sp[:, :, :C], yh, yw, C, f"i{step + 1}")

            res = self._graphs.run((f"idec_{step + 1}",) + key, after)
            if step == 3:
                x_hat = res[1]
            else:
                y_prev, sp, idx = res
                means = sp[:, :, C:]
        return {"x_hat": self._picture_out(x_hat)}


DMCI.encode_one_frame = DMCI.compress
DMCI.decode_one_frame = DMCI.decompress
