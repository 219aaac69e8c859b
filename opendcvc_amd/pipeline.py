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
    def __init__(self, i_net, p_net, qp_i, qp_p=None, intra_period=-1, reset_interval=32):
        self.i_net, self.p_net = i_net, p_net
        self.qp_i = qp_i
        self.qp_p = qp_i if qp_p is None else qp_p
        self.intra_period = intra_period
        self.reset_interval = reset_interval
        self.frame_idx = 0
        self.last_qp = 0
        p_net.set_curr_poc(0)

    def encode(self, x_padded):
        fi = self.frame_idx
        self.frame_idx += 1
        if fi == 0 or (self.intra_period > 0 and fi % self.intra_period == 0):
            enc = self.i_net.compress(x_padded, self.qp_i)
            self.p_net.clear_dpb()
            self.p_net.add_ref_frame(None, enc["x_hat"])
            return FramePacket(True, self.qp_i, 0, enc["bit_stream"])
        use_ada_i = 0
        if self.reset_interval > 0 and fi % self.reset_interval == 1:
            use_ada_i = 1
            self.p_net.prepare_feature_adaptor_i(self.last_qp)
        qp = self.p_net.shift_qp(self.qp_p, INDEX_MAP[fi % 8])
        enc = self.p_net.compress(x_padded, qp)
        self.last_qp = qp
        return FramePacket(False, qp, use_ada_i, enc["bit_stream"])


class SequenceDecoder:
    def __init__(self, i_net, p_net, height, width, use_two):
        self.i_net, self.p_net = i_net, p_net
        self.h, self.w, self.two = height, width, use_two
        p_net.set_curr_poc(0)

    def decode(self, pkt):
        sps = dict(height=self.h, width=self.w, ec_part=1 if self.two else 0, use_ada_i=pkt.use_ada_i)
        if pkt.is_i:
            dec = self.i_net.decompress(pkt.bit_stream, sps, pkt.qp)
            self.p_net.clear_dpb()
            self.p_net.add_ref_frame(None, dec["x_hat"])
        else:
            if pkt.use_ada_i:
                self.p_net.reset_ref_feature()
            dec = self.p_net.decompress(pkt.bit_stream, sps, pkt.qp)
        return dec["x_hat"]


def use_two_entropy_coders(height, width):
    """test_video.py:152"""
    return height * width > 1280 * 720
