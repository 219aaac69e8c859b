"""Per-sequence encode / decode drivers: the frame-type, qp-offset and feature-refresh policy of the
reference harness (test_video.py:164-214 encoder side, :258-285 decoder side) packaged as two
small state machines, so bench.py / smoke() / the tests all drive the codecs the same way the
reference's run_one_point_with_stream does.
"""
from dataclasses import dataclass

INDEX_MAP = (0, 1, 0, 2, 0, 2, 0, 2)          # test_video.py:164


@dataclass
class FramePacket:
    is_i: bool
    qp: int
    use_ada_i: int
    bit_stream: bytes


class SequenceEncoder:
    """defer_stream=False: encode(x) returns the frame's packet (the reference's loop).
    defer_stream=True: P-frame packets come out one call late - encode(x) returns a LIST of the packets completed by
    the call, in order, flush() the rest; the host entropy coding of a P frame then runs underneath the next frame's
    kernels (DMC.compress(defer_stream=True)), which makes a sequential encoder GPU-bound."""

    def __init__(self, i_net, p_net, qp_i, qp_p=None, intra_period=-1, reset_interval=32, defer_stream=False):
        self.i_net, self.p_net = i_net, p_net
        self.defer = defer_stream
        self._held = None            # (qp, use_ada_i) of the P frame whose stream is still pending
        self.qp_i = qp_i
        self.qp_p = qp_i if qp_p is None else qp_p
        self.intra_period = intra_period
        self.reset_interval = reset_interval
        self.frame_idx = 0
        self.last_qp = 0
        p_net.set_curr_poc(0)

    def encode(self, x_padded):
        from .models import CAPTURE_GUARD
        with CAPTURE_GUARD.frame():          # (a HIP graph capture on another thread waits for / holds back this frame)
            return self._encode(x_padded)

    def _take_held(self, stream):
        out = []
        if self._held is not None and stream is not None:
            out.append(FramePacket(False, self._held[0], self._held[1], stream))
            self._held = None
        return out

    def _encode(self, x_padded):
        fi = self.frame_idx
        self.frame_idx += 1
        if fi == 0 or (self.intra_period > 0 and fi % self.intra_period == 0):
            done = self._take_held(self.p_net.finish_stream()) if self.defer else []
            enc = self.i_net.compress(x_padded, self.qp_i)
            self.p_net.clear_dpb()
            self.p_net.add_ref_frame(None, enc["x_hat"])
            pkt = FramePacket(True, self.qp_i, 0, enc["bit_stream"])
            return done + [pkt] if self.defer else pkt
        use_ada_i = 0
        if self.reset_interval > 0 and fi % self.reset_interval == 1:
            use_ada_i = 1
            self.p_net.prepare_feature_adaptor_i(self.last_qp)
        qp = self.p_net.shift_qp(self.qp_p, INDEX_MAP[fi % 8])
        enc = self.p_net.compress(x_padded, qp, defer_stream=self.defer)
        self.last_qp = qp
        if not self.defer:
            return FramePacket(False, qp, use_ada_i, enc["bit_stream"])
        done = self._take_held(enc.get("bit_stream_prev"))
        self._held = (qp, use_ada_i)
        return done

    def flush(self):
        """defer_stream: the packet still pending (empty list otherwise)"""
        from .models import CAPTURE_GUARD
        with CAPTURE_GUARD.frame():
            return self._take_held(self.p_net.finish_stream()) if self.defer else []


class SequenceDecoder:
    """defer_output=False: decode(pkt) returns the packet's picture (the reference's loop).
    defer_output=True: P pictures come out one call late - decode(pkt) returns a LIST of the pictures completed
    by the call, in order (usually one: the previous frame), flush() the rest; the reconstruction network of a
    P frame then runs inside the host entropy-decoding gaps of the next frame (DMC.decompress)."""

    def __init__(self, i_net, p_net, height, width, use_two, defer_output=False):
        self.i_net, self.p_net = i_net, p_net
        self.h, self.w, self.two = height, width, use_two
        self.defer = defer_output
        p_net.set_curr_poc(0)

    def decode(self, pkt):
        from .models import CAPTURE_GUARD
        with CAPTURE_GUARD.frame():
            return self._decode(pkt)

    def _decode(self, pkt):
        sps = dict(height=self.h, width=self.w, ec_part=1 if self.two else 0, use_ada_i=pkt.use_ada_i)
        done = []
        if pkt.is_i:
            if self.defer:
                last = self.p_net.finish_output()
                if last is not None:
                    done.append(last)
            dec = self.i_net.decompress(pkt.bit_stream, sps, pkt.qp)
            self.p_net.clear_dpb()
            self.p_net.add_ref_frame(None, dec["x_hat"])
            done.append(dec["x_hat"])
        else:
            if pkt.use_ada_i:
                self.p_net.reset_ref_feature()
            dec = self.p_net.decompress(pkt.bit_stream, sps, pkt.qp, defer_output=self.defer)
            for k in ("x_hat_prev", "x_hat"):
                if dec.get(k) is not None:
                    done.append(dec[k])
        return done if self.defer else done[-1]

    def flush(self):
        from .models import CAPTURE_GUARD
        with CAPTURE_GUARD.frame():
            last = self.p_net.finish_output() if self.defer else None
        return [] if last is None else [last]


class EncodeDecodePipeline:
    """Encoder and decoder of one stream as a two-stage pipeline on one GPU: each stage has its own host
    thread and HIP stream, so frame n is decoded while frame n+1 is encoded and one stage's host entropy
    coding overlaps the other stage's kernels.  (The reference runs the two loops one after the other,
    test_video.py:164-214 then :258-285; the frames, packets and reconstructions are the same.)"""

    def __init__(self, encoder, decoder, device, depth=None):
        import os
        import torch
        if depth is None:
            depth = int(os.environ.get("DCVC_PIPE_DEPTH", "4"))
        self.encoder, self.decoder, self.device, self.depth = encoder, decoder, device, depth
        # (a high-priority decoder stream was measured: no difference - the pair is GPU-bound either way)
        self.enc_stream, self.dec_stream = torch.cuda.Stream(device), torch.cuda.Stream(device)

    @staticmethod
    def _emit(frames, on_frame):
        if on_frame is not None:
            for x in (frames if isinstance(frames, list) else [frames]):
                on_frame(x)

    def run(self, frames, on_packet=None, on_frame=None):
        """frames: iterable of padded model inputs (device tensors, ready on the calling stream).
        on_packet(pkt) is called on the encoder thread, on_frame(x_hat) on the decoder thread with the
        decoder stream current.  Returns when every frame has been encoded and decoded."""
        import queue
        import threading
        import torch
        from .models import CAPTURE_GUARD
        torch.cuda.current_stream().synchronize()
        frames = iter(frames)
        # HIP graph captures (first frames, new resolutions, new variants) are serialised against the other
        # stage's host calls by models.CAPTURE_GUARD inside encode() / decode(): no warm-up phase is needed.
        q = queue.Queue(maxsize=self.depth)
        errors = []

        def enc_stage():
            try:
                torch.cuda.set_device(self.device)
                with torch.cuda.stream(self.enc_stream):
                    def emit(pkts):
                        # a deferring encoder (SequenceEncoder(defer_stream=True)) returns a LIST of packets - the previous
                        # frame's, possibly none - and keeps the last one until flush()
                        for pkt in (pkts if isinstance(pkts, (list, tuple)) else [pkts]):
                            if on_packet is not None:
                                with CAPTURE_GUARD.frame():     # the callback may touch the device too
                                    on_packet(pkt)
                            q.put(pkt)                          # (never block on the queue inside the scope)

                    for x in frames:
                        with CAPTURE_GUARD.frame():
                            pkts = self.encoder.encode(x)
                        emit(pkts)
                    if getattr(self.encoder, "defer", False):
                        with CAPTURE_GUARD.frame():
                            pkts = self.encoder.flush()
                        emit(pkts)
                    with CAPTURE_GUARD.frame():
                        self.enc_stream.synchronize()
            except BaseException as e:                      # re-raised by run()
                errors.append(e)
            finally:
                q.put(None)

        def dec_stage():
            try:
                torch.cuda.set_device(self.device)
                with torch.cuda.stream(self.dec_stream):
                    while True:
                        pkt = q.get()
                        if pkt is None:
                            break
                        with CAPTURE_GUARD.frame():
                            self._emit(self.decoder.decode(pkt), on_frame)
                    with CAPTURE_GUARD.frame():
                        self._emit(self.decoder.flush(), on_frame)
                        self.dec_stream.synchronize()
            except BaseException as e:
                errors.append(e)
                while q.get() is not None:                  # keep the encoder from blocking on a full queue
                    pass

        # The two stage threads share the interpreter lock: with the default 5 ms switch interval a stage that becomes
        # runnable (its GPU event fired, its packet arrived) may wait that long for the other stage's Python code; 0.2 ms
        # for the duration of the run (steadier frame times: profiles/r04_host_threads_ab.txt)
        import sys
        old_interval = sys.getswitchinterval()
        sys.setswitchinterval(min(old_interval, 2e-4))
        threads = [threading.Thread(target=enc_stage), threading.Thread(target=dec_stage)]
        try:
            for t in threads:
                t.start()
            for t in threads:
                t.join()
        finally:
            sys.setswitchinterval(old_interval)
        if errors:
            raise errors[0]


def load_yuv420_frame(y, u, v, dtype, pad_to=16):
    """uint8 CUDA planes y [H,W], u/v [H/2,W/2] -> padded model input [1,3,H',W'] (one fused kernel;
    reference: get_src_frame + replicate_pad, test_video.py:74-91,150,179)."""
    import ctypes
    import torch
    from . import _lib
    from . import nn as L
    H, W = y.shape
    pr, pb = (-W) % pad_to, (-H) % pad_to
    out = torch.empty((1, 3, H + pb, W + pr), dtype=dtype, device=y.device)
    _lib.check(_lib.lib().dcvc_yuv420_to_frame(L.dtype_code(dtype), L._p(y.contiguous()), L._p(u.contiguous()),
                                               L._p(v.contiguous()), H, W, pb, pr, L._p(out),
                                               ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
               "dcvc_yuv420_to_frame")
    return out


def store_yuv420_frame(x_hat, height, width, round_uv=False):
    """decoded [1,3,H',W'] -> uint8 CUDA planes (y, u, v) of the height x width picture
    (reference: yuv_444_to_420 + clamp*255 + uint8, test_video.py:307-311)."""
    import ctypes
    import torch
    from . import _lib
    from . import nn as L
    x = x_hat.contiguous()
    _, _, Hp, Wp = x.shape
    y = torch.empty((height, width), dtype=torch.uint8, device=x.device)
    u = torch.empty((height // 2, width // 2), dtype=torch.uint8, device=x.device)
    v = torch.empty_like(u)
    _lib.check(_lib.lib().dcvc_frame_to_yuv420(L.dtype_code(x.dtype), L._p(x), Hp, Wp, height, width, int(round_uv),
                                               L._p(y), L._p(u), L._p(v),
                                               ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
               "dcvc_frame_to_yuv420")
    return y, u, v


def use_two_entropy_coders(height, width):
    """test_video.py:152"""
    return height * width > 1280 * 720
