"""Full DMCI / DMC codecs on the HIP path (through the model API = drop-in seam 1) against
(a) the CPU oracle on the same seeded weights and frames: fp32 mode must be BIT-EXACT (streams,
    features, reconstructions);
(b) the reference's golden records (tests/golden/sequences.json), with the fp32 tolerance of
    BASELINE.json (PSNR / bpp within 1e-4);
(c) fp16 mode: encoder/decoder self-consistency and rate/distortion close to the fp32 path."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

import dcvc_oracle as O
from opendcvc_amd import weights
from seq_utils import check_against_record, psnr_of, run_sequence

pytestmark = pytest.mark.gpu


def hip_codecs(seed, thres, dtype):
    from opendcvc_amd.models import DMC, DMCI
    i_net, p_net = DMCI(), DMC()
    i_net.load_state_dict({k: torch.from_numpy(v) for k, v in weights.make_state_dict("dmci", seed).items()})
    p_net.load_state_dict({k: torch.from_numpy(v) for k, v in weights.make_state_dict("dmc", seed).items()})
    for m in (i_net, p_net):
        m.to("cuda").eval()
        m.update(thres)
        if dtype == torch.float16:
            m.half()
    return i_net, p_net


def run_hip(rec, dtype, n_frames=None):
    i_net, p_net = hip_codecs(rec["seed"], rec["thres"], dtype)
    return run_sequence(i_net, p_net, rec, n_frames,
                        to_x=lambda a: torch.from_numpy(a).to("cuda", dtype),
                        to_np=lambda t: t.float().cpu().numpy(),
                        feature_of=lambda p: p.dpb[0].feature.float().cpu().numpy())


def run_oracle(rec, n_frames=None):
    i_net = O.OracleDMCI(weights.make_state_dict("dmci", rec["seed"]))
    p_net = O.OracleDMC(weights.make_state_dict("dmc", rec["seed"]))
    i_net.update(rec["thres"])
    p_net.update(rec["thres"])
    return run_sequence(i_net, p_net, rec, n_frames, feature_of=lambda p: p.ref_feature)


@pytest.fixture(scope="module")
def seqs(golden_dir):
    d = json.load(open(os.path.join(golden_dir, "sequences.json")))
    d["seq_64"] = json.loads(str(np.load(os.path.join(golden_dir, "seq_64.npz"))["meta"]))
    return d


@pytest.mark.parametrize("name", ["seq_64", "seq_64_two", "seq_80x48", "seq_256"])
def test_fp32_bit_exact_with_oracle_and_within_tolerance_of_reference(seqs, name):
    rec = seqs[name]
    got = run_hip(rec, torch.float32)
    ora = run_oracle(rec)
    for fi, (g, o) in enumerate(zip(got, ora)):
        assert g["bits"] == o["bits"], f"frame {fi}: HIP fp32 stream differs from the oracle"
        assert np.array_equal(g["x_hat"], o["x_hat"]), f"frame {fi}: reconstruction differs from the oracle"
        if fi > 0:
            assert np.array_equal(g["feature"], o["feature"]), f"frame {fi}: encoder feature differs"
            assert np.array_equal(g["dec_feature"], g["feature"]), f"frame {fi}: enc/dec feature desync"
    check_against_record(rec, got, min_exact=1.0 if name != "seq_256" else 0.75)


def test_fp16_self_consistent_and_close_to_fp32(seqs):
    rec = seqs["seq_256"]
    got = run_hip(rec, torch.float16)
    devs = []
    for fi, g in enumerate(got):
        if fi > 0:   # decoder reproduces the encoder's reference feature bit for bit
            assert np.array_equal(g["dec_feature"], g["feature"]), f"frame {fi}: fp16 enc/dec desync"
        f = rec["frames"][fi]
        # fp16 storage of every activation, against the fp32 reference run: PSNR within 0.05 dB; a frame's size within
        # 4 % (the two runs quantise slightly different latents from the second frame on, and a ~2 KB frame of this
        # random-weight model moves by 1-2.5 % with the compiler's instruction selection alone), the sequence's within 2 %
        # - the bounds of test_qp_sweep_matches_reference_rd_points
        # 2 % on frames of 4 KB and more, 4 % below (there one flipped symbol is already ~0.1 %); the measured per-frame
        # deviations are written to gpurun_out/f16_frame_rate_dev.json so that the bound stays justified by data
        dev = (len(g["bits"]) - f["bytes"]) / f["bytes"]
        devs.append(dict(frame=fi, bytes_ref=f["bytes"], bytes=len(g["bits"]), rel=round(dev, 5)))
        assert abs(dev) <= (0.02 if f["bytes"] >= 4096 else 0.04), (fi, len(g["bits"]), f["bytes"])
        assert abs(psnr_of(rec, fi, g["x_hat"]) - f["psnr"]) < 0.05
    total, want = sum(len(g["bits"]) for g in got), sum(f["bytes"] for f in rec["frames"])
    import json
    import os
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    json.dump(dict(frames=devs, total_rel=round((total - want) / want, 5)), open(os.path.join(out, "f16_frame_rate_dev.json"), "w"))
    assert abs(total - want) <= 0.02 * want, (total, want)


def test_fp32_1080p_matches_reference_record(seqs):
    """BASELINE configs[1] size (1088x1920 padded, two coders): the HIP fp32 path against the record the reference
    itself produced (tests/golden/make_golden.py): per-frame stream length and PSNR within 1e-4 (I, P, P).
    (1.0 M symbols per frame: a few symbols sit on a rounding boundary of the fp32 summation order, so the
    streams are not byte-identical - the same holds for the CPU oracle, tests/test_oracle_nn.py.)"""
    rec = seqs["seq_1088x1920"]
    got = run_hip(rec, torch.float32)
    check_against_record(rec, got, min_exact=0.0, tol=1e-4)
    for fi, g in enumerate(got):
        if fi > 0:
            assert np.array_equal(g["dec_feature"], g["feature"]), f"frame {fi}: enc/dec feature desync"


def test_fp16_1080p_close_to_reference_record(seqs):
    """the benchmarked mode at the benchmarked size: rate within 0.5 %, PSNR within 0.005 dB of the reference's
    fp32 record (measured: +0.06 % / +0.18 % / -0.16 %, 3e-4 dB; round 3 accepted 2 % / 0.05 dB), decoder features
    bit-identical to the encoder's"""
    rec = seqs["seq_1088x1920"]
    got = run_hip(rec, torch.float16)
    for fi, g in enumerate(got):
        f = rec["frames"][fi]
        if fi > 0:
            assert np.array_equal(g["dec_feature"], g["feature"]), f"frame {fi}: fp16 enc/dec desync"
        assert abs(len(g["bits"]) - f["bytes"]) <= 0.005 * f["bytes"], (fi, len(g["bits"]), f["bytes"])
        assert abs(psnr_of(rec, fi, g["x_hat"]) - f["psnr"]) < 0.005, (fi, psnr_of(rec, fi, g["x_hat"]), f["psnr"])


def test_fp16_1080p_against_the_references_own_fp16_run(golden_dir):
    """The benchmarked mode at the benchmarked size against the REFERENCE's own fp16 arithmetic: its DMCI / DMC in .half() on the
    CPU (tests/golden/make_golden_1080p_f16.py -> seq_1080p_f16.json; the reference's fp16 run differs from its fp32 run by
    <= 0.03 % in bytes and 2e-4 dB).  The HIP fp16 kernels round less often than the reference's, so streams
    are not byte-identical: per frame bytes within 0.5 %, PSNR within 0.005 dB (measured on MI355X: +0.07 % / +0.21 % / -0.15 %,
    |dPSNR| <= 3.2e-4 dB - written to gpurun_out/f16_1080p_vs_ref_f16.json, kept as profiles/r04_f16_1080p_vs_ref_f16.json)."""
    rec = json.load(open(os.path.join(golden_dir, "seq_1080p_f16.json")))
    got = run_hip(rec, torch.float16)
    devs = []
    for fi, g in enumerate(got):
        f = rec["frames"][fi]
        psnr = psnr_of(rec, fi, g["x_hat"])
        devs.append(dict(frame=fi, type=f["type"], bytes_ref=f["bytes"], bytes=len(g["bits"]),
                         rel=round((len(g["bits"]) - f["bytes"]) / f["bytes"], 6), psnr_ref=f["psnr"], psnr=psnr,
                         identical=hashlib.sha256(g["bits"]).hexdigest() == f["sha256"]))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    json.dump(dict(frames=devs), open(os.path.join(out, "f16_1080p_vs_ref_f16.json"), "w"), indent=1)
    for d in devs:
        assert abs(d["rel"]) <= 0.005, d
        assert abs(d["psnr"] - d["psnr_ref"]) < 0.005, d


def test_requires_update_and_cuda():
    from opendcvc_amd._lib import DcvcError
    from opendcvc_amd.models import DMCI
    m = DMCI()
    with pytest.raises(DcvcError):
        m.compress(torch.zeros(1, 3, 64, 64), 0)          # cpu model: no fallback
    m.to("cuda")
    with pytest.raises(DcvcError):
        m.compress(torch.zeros(1, 3, 64, 64, device="cuda"), 0)   # update() not called


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_deferred_decoder_output_equals_immediate(dtype):
    """SequenceDecoder(defer_output=True): the reconstruction network of a P frame runs inside the next frame's
    host-decoding gaps and the picture comes out one call late - same pictures, same order, across I frames
    (intra period 4) and feature refreshes (reset interval 3)."""
    from opendcvc_amd.pipeline import SequenceDecoder, SequenceEncoder
    h, w, n = 96, 160, 9
    frames = [torch.from_numpy(weights.synthetic_frame_yuv444(h, w, fi, 7)).to("cuda", dtype) for fi in range(n)]
    ie, pe = hip_codecs(1234, 0.12, dtype)
    for m in (ie, pe):
        m.set_use_two_entropy_coders(False)
    enc = SequenceEncoder(ie, pe, 28, intra_period=4, reset_interval=3)
    pkts = [enc.encode(x) for x in frames]
    outs = []
    for defer in (False, True):
        idc, pdc = hip_codecs(1234, 0.12, dtype)
        for m in (idc, pdc):
            m.set_use_two_entropy_coders(False)
        dec = SequenceDecoder(idc, pdc, h, w, False, defer_output=defer)
        got = []
        for pkt in pkts:
            r = dec.decode(pkt)
            got += [t.float().cpu().numpy() for t in (r if defer else [r])]
        got += [t.float().cpu().numpy() for t in dec.flush()]
        outs.append(got)
    assert len(outs[0]) == len(outs[1]) == n
    for fi, (a, b) in enumerate(zip(*outs)):
        assert np.array_equal(a, b), f"frame {fi}: deferred output differs"


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_two_stage_pipeline_equals_sequential(dtype):
    """EncodeDecodePipeline (encoder / decoder on two host threads and HIP streams) = the plain
    encode-then-decode loop: same packets, same reconstructions, frame for frame."""
    from opendcvc_amd.pipeline import EncodeDecodePipeline, SequenceDecoder, SequenceEncoder
    h, w, n = 144, 256, 7
    frames = [torch.from_numpy(weights.synthetic_frame_yuv444(h, w, fi, 3)).to("cuda", dtype) for fi in range(n)]

    def codecs():
        (ie, pe), (idc, pdc) = hip_codecs(1234, 0.12, dtype), hip_codecs(1234, 0.12, dtype)
        for m in (ie, pe, idc, pdc):
            m.set_use_two_entropy_coders(True)
        return (SequenceEncoder(ie, pe, 20, intra_period=4, reset_interval=3),
                SequenceDecoder(idc, pdc, h, w, True))

    enc, dec = codecs()
    want = []
    for x in frames:
        pkt = enc.encode(x)
        want.append((pkt, dec.decode(pkt).float().cpu().numpy()))
    enc, dec = codecs()
    dec.defer = True                      # the pipeline's usual configuration: decoder output one frame late
    pkts, outs = [], []
    EncodeDecodePipeline(enc, dec, torch.device("cuda", 0)).run(
        frames, on_packet=pkts.append, on_frame=lambda t: outs.append(t.float().cpu().numpy()))
    assert len(pkts) == len(outs) == n
    for fi, ((wp, wx), gp, gx) in enumerate(zip(want, pkts, outs)):
        assert (gp.is_i, gp.qp, gp.use_ada_i) == (wp.is_i, wp.qp, wp.use_ada_i)
        assert gp.bit_stream == wp.bit_stream, f"frame {fi}: packet differs"
        assert np.array_equal(gx, wx), f"frame {fi}: reconstruction differs"


def test_graph_replay_equals_plain_launches():
    """Captured runs (HIP graphs) vs the same kernels launched one by one: same packets and reconstructions,
    including the encoder's run-ahead of the next frame's feature extractor (graphs only)."""
    from opendcvc_amd.pipeline import SequenceDecoder, SequenceEncoder
    h, w, n = 96, 160, 6
    frames = [torch.from_numpy(weights.synthetic_frame_yuv444(h, w, fi, 5)).to("cuda", torch.float16) for fi in range(n)]
    results = []
    for graphs in (True, False):
        (ie, pe), (idc, pdc) = hip_codecs(1234, 0.12, torch.float16), hip_codecs(1234, 0.12, torch.float16)
        for m in (ie, pe, idc, pdc):
            m.set_use_two_entropy_coders(False)
            m._graphs.enabled = graphs
        enc = SequenceEncoder(ie, pe, 24, intra_period=-1, reset_interval=4)
        dec = SequenceDecoder(idc, pdc, h, w, False)
        out = []
        for x in frames:
            pkt = enc.encode(x)
            out.append((pkt.bit_stream, dec.decode(pkt).float().cpu().numpy()))
        if graphs:
            assert pe._graphs.variants("enc_front_ahead") == {"p"} and pdc._graphs.variants("dec_4") == {"i", "p"}
            assert pe._graphs.variants("enc_back") == {"full", "ahead"}
        results.append(out)
    for fi, ((b0, x0), (b1, x1)) in enumerate(zip(*results)):
        assert b0 == b1, f"frame {fi}: packet differs between graph replay and plain launches"
        assert np.array_equal(x0, x1), f"frame {fi}: reconstruction differs"


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
@pytest.mark.parametrize("two", [False, True])
def test_compacted_decoder_hand_off_equals_whole_array_hand_off(dtype, two):
    """The decoder with its hand-off compacted on the device (default) against the whole-array copies of round 3
    (models.DEC_COMPACT = False, what DCVC_DEC_COMPACT=0 selects): the same pictures from the same packets, I frame (four
    steps) and P frames (two steps), a map whose pixel count is not a multiple of 16 (80 x 112 -> 5 x 7 latents), one and two
    coders."""
    from opendcvc_amd import models
    from opendcvc_amd.pipeline import SequenceDecoder, SequenceEncoder
    h, w, n = 80, 112, 4
    frames = [torch.from_numpy(weights.synthetic_frame_yuv444(h, w, fi, 9)).to("cuda", dtype) for fi in range(n)]
    ie, pe = hip_codecs(1234, 0.12, dtype)
    for m in (ie, pe):
        m.set_use_two_entropy_coders(two)
    enc = SequenceEncoder(ie, pe, 40, intra_period=-1, reset_interval=32)
    pkts = [enc.encode(x) for x in frames]
    outs = []
    old = models.DEC_COMPACT
    try:
        for compact in (True, False):
            models.DEC_COMPACT = compact
            idc, pdc = hip_codecs(1234, 0.12, dtype)
            dec = SequenceDecoder(idc, pdc, h, w, two)
            outs.append([dec.decode(p).float().cpu().numpy() for p in pkts])
            keys = [k for k in pdc.entropy_coder._pinned if isinstance(k[0], str)]
            assert any(k[0].endswith("_cidx") for k in keys) == compact and any(k[0].endswith("_idx") for k in keys) == (not compact)
    finally:
        models.DEC_COMPACT = old
    for fi, (a, b) in enumerate(zip(*outs)):
        assert np.array_equal(a, b), f"frame {fi}: the two hand-off forms decode different pictures"


@pytest.mark.parametrize("name", ["edge_q0", "edge_q63", "edge_nothres", "edge_allskip"])
def test_fp32_edge_case_sequences_match_reference_records(golden_dir, name):
    """the same edge cases against the records the reference itself produced (sequences_edge.json)"""
    rec = json.load(open(os.path.join(golden_dir, "sequences_edge.json")))[name]
    check_against_record(rec, run_hip(rec, torch.float32), min_exact=1.0 if name != "edge_q0" else 0.6)


@pytest.mark.parametrize("qp,thres,two,hw", [(0, 0.12, 0, (64, 64)), (63, 0.12, 0, (64, 64)), (32, None, 1, (64, 64)),
                                             (32, 100.0, 1, (64, 64)), (17, 0.12, 0, (48, 112))])
def test_fp32_edge_cases_bit_exact_with_oracle(qp, thres, two, hw):
    """qp at both ends of the table (P frames reach the extra entries 64..71), no force-zero threshold, a threshold
    that skips every y symbol (the y part of the stream is empty), a non-square map: HIP fp32 == oracle, streams
    and reconstructions, I + 2 P frames, separate encoder-side and decoder-side models."""
    h, w = hw
    hip = [hip_codecs(1234, thres, torch.float32) for _ in range(2)]          # [(i, p) encoder side, (i, p) decoder side]
    ora = []
    for _ in range(2):
        i_o, p_o = O.OracleDMCI(weights.make_state_dict("dmci", 1234)), O.OracleDMC(weights.make_state_dict("dmc", 1234))
        i_o.update(thres)
        p_o.update(thres)
        ora.append((i_o, p_o))
    for pair in hip + ora:
        for m in pair:
            m.set_use_two_entropy_coders(bool(two))
    sps = dict(height=h, width=w, ec_part=two, use_ada_i=0)
    for fi in range(3):
        x = weights.synthetic_frame_yuv444(h, w, fi, 0)
        xd = torch.from_numpy(x).cuda()
        if fi == 0:
            eh, eo = hip[0][0].compress(xd, qp), ora[0][0].compress(x, qp)
            cur = qp
        else:
            cur = hip[0][1].shift_qp(qp, [0, 1, 0, 2][fi % 4])
            eh, eo = hip[0][1].compress(xd, cur), ora[0][1].compress(x, cur)
        assert eh["bit_stream"] == eo["bit_stream"], f"frame {fi}: stream differs from the oracle"
        if thres is not None and thres > 50 and fi > 0:
            assert len(eh["bit_stream"]) < 400                                # only z survives
        net = 0 if fi == 0 else 1
        dh = hip[1][net].decompress(eh["bit_stream"], sps, cur)
        do = ora[1][net].decompress(eo["bit_stream"], sps, cur)
        assert np.array_equal(dh["x_hat"].cpu().numpy(), np.asarray(do["x_hat"])), f"frame {fi}: reconstruction differs"
        if fi == 0:
            hip[0][1].clear_dpb(); hip[0][1].add_ref_frame(None, eh["x_hat"])
            hip[1][1].clear_dpb(); hip[1][1].add_ref_frame(None, dh["x_hat"])
            ora[0][1].clear_dpb(); ora[0][1].add_ref_frame(None, eo["x_hat"])
            ora[1][1].clear_dpb(); ora[1][1].add_ref_frame(None, do["x_hat"])


def _encode_decode(codecs_enc, codecs_dec, h, w, n, dtype, qp=30, seed=11):
    """I + (n-1) P frames through given encoder-side / decoder-side (DMCI, DMC) pairs"""
    from opendcvc_amd.pipeline import SequenceDecoder, SequenceEncoder
    for m in codecs_enc + codecs_dec:
        m.set_use_two_entropy_coders(False)
    enc = SequenceEncoder(codecs_enc[0], codecs_enc[1], qp, intra_period=-1, reset_interval=0)
    dec = SequenceDecoder(codecs_dec[0], codecs_dec[1], h, w, False)
    out = []
    for fi in range(n):
        x = torch.from_numpy(weights.synthetic_frame_yuv444(h, w, fi, seed)).to("cuda", dtype)
        pkt = enc.encode(x)
        out.append((pkt.bit_stream, dec.decode(pkt).float().cpu().numpy()))
    return out


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
def test_one_model_instance_across_resolutions(dtype):
    """The reference harness reuses one i_frame_net / p_frame_net per worker across sequences of different sizes
    (test_video.py init_func / worker).  Captured runs keep raw pointers to pinned staging buffers: a model that
    codes 64x64, then 256x256, then 64x64 again must give the results of fresh instances every time."""
    shared_e, shared_d = hip_codecs(1234, 0.12, dtype), hip_codecs(1234, 0.12, dtype)
    for (h, w) in ((64, 64), (256, 256), (64, 64), (96, 160), (256, 256)):
        want = _encode_decode(hip_codecs(1234, 0.12, dtype), hip_codecs(1234, 0.12, dtype), h, w, 3, dtype)
        got = _encode_decode(shared_e, shared_d, h, w, 3, dtype)
        for fi, ((wb, wx), (gb, gx)) in enumerate(zip(want, got)):
            assert gb == wb, f"{h}x{w} frame {fi}: packet differs on a reused model"
            assert np.array_equal(gx, wx), f"{h}x{w} frame {fi}: reconstruction differs on a reused model"


def test_update_twice_keeps_working():
    """update() rebuilds the entropy coder (new threshold): pinned staging that captured runs point at must
    survive it, and the runs must pick up the new threshold."""
    dtype = torch.float16
    e, d = hip_codecs(1234, 0.12, dtype), hip_codecs(1234, 0.12, dtype)
    first = _encode_decode(e, d, 64, 96, 3, dtype)
    for m in e + d:
        m.update(0.3)
    other = _encode_decode(e, d, 64, 96, 3, dtype)
    fe, fd = hip_codecs(1234, 0.3, dtype), hip_codecs(1234, 0.3, dtype)
    want = _encode_decode(fe, fd, 64, 96, 3, dtype)
    assert [b for b, _ in other] == [b for b, _ in want] and all(np.array_equal(a[1], b[1]) for a, b in zip(other, want))
    assert [b for b, _ in other] != [b for b, _ in first]          # the threshold did change the streams
    for m in e + d:
        m.update(0.12)
    again = _encode_decode(e, d, 64, 96, 3, dtype)
    assert [b for b, _ in again] == [b for b, _ in first] and all(np.array_equal(a[1], b[1]) for a, b in zip(again, first))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_deferred_encoder_stream_equals_immediate(dtype):
    """SequenceEncoder(defer_stream=True): a P frame's symbols are entropy coded during the next call and its packet
    comes out one call late - same packets, same order, across I frames (intra period 4) and feature refreshes."""
    from opendcvc_amd.pipeline import SequenceEncoder
    h, w, n = 96, 160, 10
    frames = [torch.from_numpy(weights.synthetic_frame_yuv444(h, w, fi, 7)).to("cuda", dtype) for fi in range(n)]
    outs = []
    for defer in (False, True):
        ie, pe = hip_codecs(1234, 0.12, dtype)
        for m in (ie, pe):
            m.set_use_two_entropy_coders(False)
        enc = SequenceEncoder(ie, pe, 28, intra_period=4, reset_interval=3, defer_stream=defer)
        got = []
        for x in frames:
            r = enc.encode(x)
            got += r if defer else [r]
        got += enc.flush()
        outs.append(got)
    assert len(outs[0]) == len(outs[1]) == n
    for fi, (a, b) in enumerate(zip(*outs)):
        assert (a.is_i, a.qp, a.use_ada_i) == (b.is_i, b.qp, b.use_ada_i), fi
        assert a.bit_stream == b.bit_stream, f"frame {fi}: deferred stream differs"


def test_pipeline_with_deferring_encoder_emits_every_packet(seqs):
    """EncodeDecodePipeline with SequenceEncoder(defer_stream=True): encode() then returns lists (the previous frame's
    packet, possibly none) and the last packet only comes out of flush() - the pipeline forwards every packet, in order,
    and the decoded pictures equal the plain sequential run's."""
    from opendcvc_amd.pipeline import EncodeDecodePipeline, SequenceDecoder, SequenceEncoder
    rec = seqs["seq_64_two"] if "seq_64_two" in seqs else seqs["seq_256"]
    h, w, qp = rec["h"], rec["w"], rec["qp"]
    n = min(5, len(rec["frames"]))
    frames = [torch.from_numpy(weights.synthetic_frame_yuv444(h, w, fi, 0)).cuda().half() for fi in range(n)]
    outs = {}
    for defer in (False, True):
        (ie, pe), (idec, pdec) = hip_codecs(rec["seed"], rec["thres"], torch.float16), hip_codecs(rec["seed"], rec["thres"], torch.float16)
        for m in (ie, pe, idec, pdec):
            m.set_use_two_entropy_coders(bool(rec["two"]))
        enc = SequenceEncoder(ie, pe, qp, intra_period=-1, reset_interval=0, defer_stream=defer)
        dec = SequenceDecoder(idec, pdec, h, w, bool(rec["two"]), defer_output=True)
        pkts, pics = [], []
        EncodeDecodePipeline(enc, dec, torch.device("cuda", 0)).run(frames, lambda p: pkts.append(p.bit_stream),
                                                                   lambda x: pics.append(x.float().cpu().numpy()))
        outs[defer] = (pkts, pics)
    assert len(outs[True][0]) == len(outs[False][0]) == n and outs[True][0] == outs[False][0]
    assert len(outs[True][1]) == n and all(np.array_equal(a, b) for a, b in zip(outs[True][1], outs[False][1]))


@pytest.mark.parametrize("defer", [False, True])
def test_returned_pictures_stay_valid_when_kept(defer):
    """The reference returns a fresh x_hat per frame and its callers keep them (the harness computes a sequence's PSNR
    afterwards): every picture handed out must still hold its own frame after later frames have been decoded.  (Round 3
    wrote pictures into two alternating buffers: the third kept picture aliased the first - ADVICE round 3.)"""
    from opendcvc_amd.pipeline import SequenceDecoder, SequenceEncoder
    h, w, n = 96, 160, 7
    dtype = torch.float16
    frames = [torch.from_numpy(weights.synthetic_frame_yuv444(h, w, fi, 11)).to("cuda", dtype) for fi in range(n)]
    ie, pe = hip_codecs(1234, 0.12, dtype)
    idc, pdc = hip_codecs(1234, 0.12, dtype)
    for m in (ie, pe, idc, pdc):
        m.set_use_two_entropy_coders(False)
    enc = SequenceEncoder(ie, pe, 28, intra_period=-1, reset_interval=32)
    dec = SequenceDecoder(idc, pdc, h, w, False, defer_output=defer)
    kept, copies = [], []
    for x in frames:
        r = dec.decode(enc.encode(x))
        for t in (r if defer else [r]):
            kept.append(t)                                  # the tensor itself ...
            copies.append(t.float().cpu().numpy().copy())   # ... and its content at the time it was handed out
    for t in dec.flush():
        kept.append(t)
        copies.append(t.float().cpu().numpy().copy())
    assert len(kept) == n and len({t.data_ptr() for t in kept}) == n
    for fi, (t, c) in enumerate(zip(kept, copies)):
        assert np.array_equal(t.float().cpu().numpy(), c), f"picture {fi} was overwritten by a later frame"
    assert not np.array_equal(copies[1], copies[3])


def test_one_instance_refuses_a_second_thread():
    """one model instance = one host thread at a time (its captured runs share scratch and the branch stream): a second
    thread is refused with DcvcError instead of interleaving kernels on shared scratch"""
    import threading
    from opendcvc_amd._lib import DcvcError
    _, p_net = hip_codecs(1234, 0.12, torch.float16)
    seen = []

    def intruder():
        try:
            with p_net._frame():
                seen.append("entered")
        except DcvcError as e:
            seen.append(str(e))

    with p_net._frame():
        with p_net._frame():          # re-entrant for the owner
            t = threading.Thread(target=intruder)
            t.start()
            t.join()
    assert len(seen) == 1 and "one thread at a time" in seen[0]
    t = threading.Thread(target=intruder)     # free again afterwards
    t.start()
    t.join()
    assert seen[-1] == "entered"


def test_damaged_payload_raises_instead_of_decoding_garbage():
    """DMCI / DMC.decompress check the coder's end state after the frame's last symbol (dcvc_rans_dec_check_end): a
    truncated or bit-flipped payload raises DcvcError; the models stay usable (the reference decodes garbage,
    rans.cpp:356-429)."""
    from opendcvc_amd._lib import DcvcError
    h, w = 96, 160
    dtype = torch.float16
    frames = [torch.from_numpy(weights.synthetic_frame_yuv444(h, w, fi, 5)).to("cuda", dtype) for fi in range(2)]
    ie, pe = hip_codecs(1234, 0.12, dtype)
    idc, pdc = hip_codecs(1234, 0.12, dtype)
    for m in (ie, pe, idc, pdc):
        m.set_use_two_entropy_coders(False)
    sps = dict(height=h, width=w, ec_part=0, use_ada_i=0)
    ei = ie.compress(frames[0], 30)
    pe.clear_dpb()
    pe.add_ref_frame(None, ei["x_hat"])
    ep = pe.compress(frames[1], 30)
    for bits in (ei["bit_stream"][:-3], ei["bit_stream"][:len(ei["bit_stream"]) // 2], ei["bit_stream"] + b"\x01"):
        with pytest.raises(DcvcError):
            idc.decompress(bits, sps, 30)
    di = idc.decompress(ei["bit_stream"], sps, 30)
    assert torch.equal(di["x_hat"], ei["x_hat"])
    flipped = bytearray(ep["bit_stream"])
    flipped[7] ^= 0x10
    for bits in (ep["bit_stream"][:-2], bytes(flipped)):
        pdc.clear_dpb()
        pdc.add_ref_frame(None, di["x_hat"])
        with pytest.raises(DcvcError):
            pdc.decompress(bits, sps, 30)
    pdc.clear_dpb()
    pdc.add_ref_frame(None, di["x_hat"])
    dp = pdc.decompress(ep["bit_stream"], sps, 30)
    assert torch.equal(pdc.dpb[0].feature, pe.dpb[0].feature) and dp["x_hat"] is not None


@pytest.mark.parametrize("mode,dtype", [("fp32", torch.float32), ("fp16", torch.float16)])
def test_1080p_nine_frames_stay_on_the_reference_record(golden_dir, mode, dtype):
    """Parity along the temporal chain at the benchmarked size: I + 8 P frames at 1088 x 1920 with a feature refresh every 4
    frames, against the REFERENCE's records of the same sequence in the same arithmetic (its fp32 run / its .half() run on the
    CPU, tests/golden/make_golden_1080p_long.py -> seq_1080p_long.json).  Bounds per frame - fp32: bytes within 0.1 %, PSNR within
    1e-3 dB; fp16: bytes within 0.5 %, PSNR within 0.005 dB; the encoder's and the decoder's features stay bit-identical.  The
    measured deviations go to gpurun_out/seq_1080p_long_<mode>.json (profiles/r04_seq_1080p_long_<mode>.json)."""
    gold = json.load(open(os.path.join(golden_dir, "seq_1080p_long.json")))
    rec = dict(h=gold["h"], w=gold["w"], qp=gold["qp"], two=gold["two"], reset_interval=gold["reset_interval"], seed=gold["seed"],
               thres=gold["thres"], frames=gold[mode])
    got = run_hip(rec, dtype)
    devs = []
    for fi, g in enumerate(got):
        f = rec["frames"][fi]
        if fi > 0:
            assert np.array_equal(g["dec_feature"], g["feature"]), f"frame {fi}: enc/dec feature desync"
        devs.append(dict(frame=fi, type=f["type"], use_ada_i=f["use_ada_i"], bytes_ref=f["bytes"], bytes=len(g["bits"]),
                         rel=round((len(g["bits"]) - f["bytes"]) / f["bytes"], 6), psnr_ref=f["psnr"],
                         psnr=psnr_of(rec, fi, g["x_hat"]), identical=hashlib.sha256(g["bits"]).hexdigest() == f["sha256"]))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    json.dump(dict(frames=devs), open(os.path.join(out, f"seq_1080p_long_{mode}.json"), "w"), indent=1)
    tol_b, tol_p = (1e-3, 1e-3) if mode == "fp32" else (5e-3, 5e-3)
    for d in devs:
        assert abs(d["rel"]) <= tol_b, d
        assert abs(d["psnr"] - d["psnr_ref"]) < tol_p, d


@pytest.mark.parametrize("mode,dtype", [("fp32", torch.float32), ("fp16", torch.float16)])
def test_4k_matches_the_reference_record(golden_dir, mode, dtype):
    """BASELINE.json configs[3]'s size: an I and a P frame at 3840 x 2160 (no padding: 2160 = 16 x 135; y 135 x 240 is padded to
    136 x 240 for the hyper path) against the REFERENCE's records in the same arithmetic (tests/golden/seq_2160p.json, its fp32
    run and its .half() run).  fp32: bytes within 0.1 %, PSNR within 1e-3 dB; fp16: 0.5 % / 0.005 dB; deviations in
    gpurun_out/seq_2160p_<mode>.json (profiles/r04_seq_2160p_<mode>.json)."""
    gold = json.load(open(os.path.join(golden_dir, "seq_2160p.json")))
    rec = dict(h=gold["h"], w=gold["w"], qp=gold["qp"], two=gold["two"], reset_interval=gold["reset_interval"], seed=gold["seed"],
               thres=gold["thres"], frames=gold[mode])
    got = run_hip(rec, dtype)
    devs = []
    for fi, g in enumerate(got):
        f = rec["frames"][fi]
        if fi > 0:
            assert np.array_equal(g["dec_feature"], g["feature"]), f"frame {fi}: enc/dec feature desync"
        devs.append(dict(frame=fi, type=f["type"], bytes_ref=f["bytes"], bytes=len(g["bits"]),
                         rel=round((len(g["bits"]) - f["bytes"]) / f["bytes"], 6), psnr_ref=f["psnr"],
                         psnr=psnr_of(rec, fi, g["x_hat"]), identical=hashlib.sha256(g["bits"]).hexdigest() == f["sha256"]))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    json.dump(dict(frames=devs), open(os.path.join(out, f"seq_2160p_{mode}.json"), "w"), indent=1)
    tol_b, tol_p = (1e-3, 1e-3) if mode == "fp32" else (5e-3, 5e-3)
    for d in devs:
        assert abs(d["rel"]) <= tol_b, d
        assert abs(d["psnr"] - d["psnr_ref"]) < tol_p, d


def test_config0_single_rgb_intra_frame_matches_reference(golden_dir):
    """BASELINE.json configs[0] to the letter: DCVC-RT-Intra on ONE 256 x 256 RGB frame at q = 32, against what the REFERENCE's DMCI
    produced on its PyTorch CPU fallback path (tests/golden/make_golden_config0.py -> config0_rgb.json; the picture is prepared by
    np_image_to_tensor + rgb2ycbcr there, by the fused dcvc_rgb_to_frame kernel here).  fp32: the same stream length (within
    1e-4), RGB PSNR within 1e-4 dB, the decoder reproduces the encoder's reconstruction bit for bit."""
    import sys
    from opendcvc_amd.harness import load_rgb_frame, rgb_distortion
    sys.path.insert(0, golden_dir)
    from make_golden_png import synthetic_rgb
    gold = json.load(open(os.path.join(golden_dir, "config0_rgb.json")))
    i_net, _ = hip_codecs(gold["seed"], gold["thres"], torch.float32)
    i_net.set_use_two_entropy_coders(False)
    rgb = torch.from_numpy(synthetic_rgb(gold["size"], gold["size"], 0, gold["src_seed"])).cuda()
    x = load_rgb_frame(rgb, torch.float32)
    enc = i_net.compress(x, gold["qp"])
    dec = i_net.decompress(enc["bit_stream"], dict(height=gold["size"], width=gold["size"], ec_part=0, use_ada_i=0), gold["qp"])
    assert torch.equal(dec["x_hat"], enc["x_hat"])
    assert abs(len(enc["bit_stream"]) - gold["bytes"]) <= max(1, 1e-4 * gold["bytes"])
    psnr, _ = rgb_distortion(dec["x_hat"], rgb)
    assert abs(psnr[0] - gold["psnr_rgb"]) < 1e-4
    # (byte identity with the reference's stream is reported, not required: one flipped symbol is within the fp32 bar)
    print("config0: stream identical to the reference's:", hashlib.sha256(enc["bit_stream"]).hexdigest() == gold["sha256"])
