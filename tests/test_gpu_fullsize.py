"""BASELINE.json's full-size configurations through size-independent properties (the oracle would take
minutes per frame here): 1080p and 4K, fp16, I frame + P frames with a feature-adaptor reset.
  * encode -> container -> decode round trip: every frame decodes, decoder reference features are
    bit-identical to the encoder's (no enc/dec desync in the prediction chain);
  * determinism: encoding the same frame twice from the same state gives the same bytes;
  * the bitstream is smaller than the raw frame by a wide margin and the two-coder flag follows the
    reference's rule (test_video.py:152)."""
import io
import os

import numpy as np
import pytest
import torch

from opendcvc_amd import bitstream, weights
from opendcvc_amd.pipeline import SequenceDecoder, SequenceEncoder, use_two_entropy_coders

pytestmark = pytest.mark.gpu


def _models():
    from opendcvc_amd.models import DMC, DMCI
    out = []
    for _ in range(2):
        i_net, p_net = DMCI(), DMC()
        i_net.load_state_dict({k: torch.from_numpy(v) for k, v in weights.make_state_dict("dmci", 1234).items()})
        p_net.load_state_dict({k: torch.from_numpy(v) for k, v in weights.make_state_dict("dmc", 1234).items()})
        for m in (i_net, p_net):
            m.to("cuda").eval()
            m.update(0.12)
            m.half()
        out.append((i_net, p_net))
    return out


@pytest.mark.parametrize("h,w,frames", [(1080, 1920, 5), (2160, 3840, 3)])
def test_round_trip_full_size(h, w, frames):
    from opendcvc_amd.models import DMCI
    (ie, pe), (idec, pdec) = _models()
    two = use_two_entropy_coders(h, w)
    assert two
    for m in (ie, pe, idec, pdec):
        m.set_use_two_entropy_coders(two)
    pr, pb = DMCI.get_padding_size(h, w, 16)
    enc = SequenceEncoder(ie, pe, 32, intra_period=-1, reset_interval=3)
    f = io.BytesIO()
    writer = bitstream.StreamWriter(f)
    enc_features, sizes = [], []
    xs = []
    for fi in range(frames):
        x = np.pad(weights.synthetic_frame_yuv444(h, w, fi, 0), ((0, 0), (0, 0), (0, pb), (0, pr)), mode="edge")
        x = torch.from_numpy(x).to("cuda", torch.float16)
        xs.append(x)
        pkt = enc.encode(x)
        sizes.append(writer.write_frame(h, w, two, pkt))
        enc_features.append(None if pkt.is_i else pe.dpb[0].feature.clone())
    assert sum(sizes) < 0.1 * frames * h * w * 1.5            # far below raw YUV420
    # decode from the container
    reader = bitstream.StreamReader(io.BytesIO(f.getvalue()))
    dec = SequenceDecoder(idec, pdec, h, w, two)
    from opendcvc_amd.pipeline import FramePacket
    for fi in range(frames):
        sps, is_i, qp, payload = reader.read_frame()
        assert (sps["height"], sps["width"], sps["ec_part"]) == (h, w, 1)
        x_hat = dec.decode(FramePacket(is_i, qp, sps["use_ada_i"], payload))
        assert x_hat.shape == (1, 3, h + pb, w + pr) and torch.isfinite(x_hat).all()
        assert float(x_hat.min()) >= 0.0 and float(x_hat.max()) <= 1.0
        if not is_i:
            assert torch.equal(pdec.dpb[0].feature, enc_features[fi]), f"frame {fi}: enc/dec feature desync"
    # determinism of the encoder from an identical state
    (ie2, pe2), _ = _models()[0], None
    for m in (ie2, pe2):
        m.set_use_two_entropy_coders(two)
    enc2 = SequenceEncoder(ie2, pe2, 32, intra_period=-1, reset_interval=3)
    f2 = io.BytesIO()
    w2 = bitstream.StreamWriter(f2)
    for x in xs:
        w2.write_frame(h, w, two, enc2.encode(x))
    assert f2.getvalue() == f.getvalue()


def test_bench_gop_matches_the_references_fp16_run(golden_dir):
    """BASELINE.json configs[1] itself: the 32-frame 1080p GOP bench.py codes (synthetic YUV 4:2:0 planes, qp 32, intra period
    32, two coders) through SequenceEncoder / SequenceDecoder in fp16, against the REFERENCE's own .half() run of the same frames
    (tests/golden/make_golden_bench_gop.py -> bench_gop_f16.json; frames prepared and scored with the reference harness's
    functions).  Per frame: bytes within 1 %, weighted PSNR within 0.01 dB (measured: 0.43 %, 5.7e-4 dB); the GOP: bpp within 0.1 %,
    mean PSNR within 5e-4 dB (measured: +0.017 %, 7e-7 dB).
    Deviations in gpurun_out/bench_gop_vs_ref.json (profiles/r04_bench_gop_vs_ref.json)."""
    import json
    from opendcvc_amd import weights
    from opendcvc_amd.harness import yuv420_distortion
    from opendcvc_amd.models import DMC, DMCI
    from opendcvc_amd.pipeline import SequenceDecoder, SequenceEncoder, load_yuv420_frame
    gold = json.load(open(os.path.join(golden_dir, "bench_gop_f16.json")))
    H, W, GOP, QP = gold["height"], gold["width"], gold["gop"], gold["qp"]

    def nets():
        out = []
        for cls, name in ((DMCI, "dmci"), (DMC, "dmc")):
            m = cls()
            m.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in weights.make_state_dict(name, gold["seed"]).items()})
            m.to("cuda").eval()
            m.update(gold["thres"])
            m.half()
            m.set_use_two_entropy_coders(True)
            out.append(m)
        return out

    (ie, pe), (idc, pdc) = nets(), nets()
    enc = SequenceEncoder(ie, pe, QP, intra_period=GOP, reset_interval=GOP)
    dec = SequenceDecoder(idc, pdc, H, W, True)
    devs, total = [], 0
    for fi in range(GOP):
        planes = [torch.from_numpy(a).cuda() for a in weights.synthetic_frame_yuv420(H, W, fi, gold["src_seed"])]
        pkt = enc.encode(load_yuv420_frame(*planes, torch.float16))
        x_hat = dec.decode(pkt)
        psnr = yuv420_distortion(x_hat, *planes)
        f = gold["frames"][fi]
        assert (pkt.is_i, pkt.qp, pkt.use_ada_i) == (f["type"] == "I", f["qp"], f["use_ada_i"])
        total += len(pkt.bit_stream)
        devs.append(dict(frame=fi, bytes_ref=f["bytes"], bytes=len(pkt.bit_stream), rel=round(len(pkt.bit_stream) / f["bytes"] - 1, 6),
                         psnr_ref=f["psnr"][0], psnr=psnr[0]))
    bpp = total * 8.0 / (GOP * H * W)
    mean_psnr = float(np.mean([d["psnr"] for d in devs]))
    res = dict(gop_bpp=bpp, gop_bpp_ref=gold["gop_bpp"], bpp_rel=bpp / gold["gop_bpp"] - 1, psnr_mean=mean_psnr,
               psnr_mean_ref=gold["psnr_mean"][0], frames=devs)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    json.dump(res, open(os.path.join(out, "bench_gop_vs_ref.json"), "w"), indent=1)
    for d in devs:
        assert abs(d["rel"]) <= 0.01, d
        assert abs(d["psnr"] - d["psnr_ref"]) < 0.01, d
    assert abs(res["bpp_rel"]) <= 1e-3 and abs(mean_psnr - gold["psnr_mean"][0]) < 5e-4, (res["bpp_rel"], mean_psnr)
