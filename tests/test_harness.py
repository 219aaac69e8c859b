"""Harness log: summarize() / calc_psnr / sweep_qps against golden vectors produced by the reference's own
generate_log_json / calc_psnr (tests/golden/make_golden_harness.py), and the one-rate-point driver on the GPU."""
import io
import json
import os

import numpy as np
import pytest

from opendcvc_amd import harness


@pytest.fixture(scope="module")
def gold(golden_dir):
    return json.load(open(os.path.join(golden_dir, "harness_log.json")))


def test_log_matches_reference_schema_and_values(gold):
    for c in gold["cases"]:
        kw = {} if c["times"] is None else dict(avg_encoding_time=c["times"][0], avg_decoding_time=c["times"][1])
        got = harness.summarize(c["frame_pixel_num"], c["test_time"], c["frame_types"], c["bits"], c["psnrs"], c["ssims"],
                                verbose=c["verbose"], **kw)
        want = c["expect"]
        assert list(got.keys()) == [k for k, _ in want], c["name"]            # same keys in the same order
        for k, v in want:
            assert np.allclose(np.asarray(got[k], np.float64), np.asarray(v, np.float64), rtol=1e-13, atol=0), (c["name"], k)
        json.dumps(got)                                                        # serialisable as is


def test_psnr_and_sweep_points(gold):
    for k in gold["psnr"]:
        assert harness.calc_psnr(np.array(k["a"], np.uint8), np.array(k["b"], np.float32)) == pytest.approx(k["psnr"], rel=1e-12)
    assert harness.psnr_from_mse(float("nan")) == -999.9 and harness.psnr_from_mse(0.0) == 99.9
    for r, q in gold["sweep_qps"].items():
        assert harness.sweep_qps(int(r)) == q
    with pytest.raises(ValueError):
        harness.sweep_qps(1)


@pytest.mark.gpu
def test_one_rate_point_end_to_end(tmp_path):
    """YUV file -> container -> decode: frame types / bits follow the container, PSNR equals a numpy
    recomputation from the decoded planes, the .bin parses with the container reader."""
    import torch
    from opendcvc_amd import weights
    from opendcvc_amd.bitstream import StreamReader
    from opendcvc_amd.models import DMC, DMCI
    H, W, N = 72, 88, 6
    rng = np.random.default_rng(3)
    src = tmp_path / "seq.yuv"
    with open(src, "wb") as f:
        for fi in range(N):
            x = weights.synthetic_frame_yuv444(H, W, fi, 9)            # [1,3,H,W] in [0,1]
            y = np.clip(np.round(x[0, 0] * 255), 0, 255).astype(np.uint8)
            uv = np.clip(np.round(x[0, 1:].reshape(2, H // 2, 2, W // 2, 2).mean((2, 4)) * 255), 0, 255).astype(np.uint8)
            f.write(y.tobytes() + uv[0].tobytes() + uv[1].tobytes())
    nets = []
    for cls, name in ((DMCI, "dmci"), (DMC, "dmc")):
        m = cls()
        m.load_state_dict({k: torch.from_numpy(v) for k, v in weights.make_state_dict(name, 1234).items()})
        m.to("cuda").eval()
        m.update(0.12)
        nets.append(m.half())
    binp, recp = tmp_path / "o.bin", tmp_path / "rec.yuv"
    log = harness.run_one_point(nets[0], nets[1], str(src), W, H, N, 30, 30, intra_period=4, reset_interval=3,
                                bin_path=str(binp), rec_path=str(recp), verbose_json=True)
    assert log["i_frame_num"] == 2 and log["p_frame_num"] == 4 and log["frame_type"] == [0, 1, 1, 1, 0, 1]
    assert sum(log["frame_bpp"]) * H * W == pytest.approx(8 * os.path.getsize(binp))
    rd = StreamReader(io.BytesIO(open(binp, "rb").read()))
    kinds = [rd.read_frame()[1] for _ in range(N)]
    assert kinds == [True, False, False, False, True, False]
    rec = np.frombuffer(open(recp, "rb").read(), np.uint8).reshape(N, H * W * 3 // 2)
    raw = np.frombuffer(open(src, "rb").read(), np.uint8).reshape(N, H * W * 3 // 2)
    for fi in range(N):       # the written planes are rounded / truncated, the logged PSNR is not: within 0.1 dB
        assert abs(harness.calc_psnr(raw[fi, :H * W], rec[fi, :H * W]) - log["frame_psnr_y"][fi]) < 0.1
    assert 0 < log["ave_all_frame_psnr"] < 60 and log["ave_all_frame_msssim"] == 0      # (untrained synthetic weights: single-digit dB)


def test_ramped_qp_tables_leave_the_other_weights_alone():
    """weights.make_state_dict(q_ramp=True) (the weights of tests/golden/sweep_ramp.json): only the per-qp tables differ from
    the default draw, they span 16:1 over qp, encoder-side tables rise and decoder-side ones fall."""
    from opendcvc_amd import weights
    for name in ("dmc", "dmci"):
        a, b = weights.make_state_dict(name, 1234), weights.make_state_dict(name, 1234, q_ramp=True)
        assert list(a) == list(b)
        changed = [k for k in a if not np.array_equal(a[k], b[k])]
        assert changed and all(k.startswith("q_") for k in changed), changed
        for k in changed:
            lo, hi = float(b[k][0].mean()), float(b[k][63].mean())
            ratio = hi / lo if not ("dec" in k or "recon" in k) else lo / hi
            assert 12.0 < ratio < 20.0, (k, lo, hi)


def _nets(dtype, q_ramp=False):
    import torch
    from opendcvc_amd import weights
    from opendcvc_amd.models import DMC, DMCI
    nets = []
    for cls, name in ((DMCI, "dmci"), (DMC, "dmc")):
        m = cls()
        m.load_state_dict({k: torch.from_numpy(v) for k, v in weights.make_state_dict(name, 1234, q_ramp=q_ramp).items()})
        m.to("cuda").eval()
        m.update(0.12)
        nets.append(m.half() if dtype == "fp16" else m)
    return nets


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["fp32", "fp16"])
@pytest.mark.parametrize("gold_name", ["sweep.json", "sweep_ramp.json"])
def test_qp_sweep_matches_reference_rd_points(tmp_path, golden_dir, mode, gold_name):
    """BASELINE.json configs[2]: the qp sweep {0, 21, 42, 63} of one YUV 4:2:0 file through harness.run_sweep against
    the log the REFERENCE's own run_one_point_with_stream wrote for the same file and weights
    (tests/golden/make_golden_sweep.py -> sweep.json): same keys in the same order; fp32: every frame's bits within
    one byte (= bpp within 1e-4 at these stream sizes is byte-exactness) and PSNR within 1e-4 dB; fp16: bpp within
    2 %, PSNR within 0.05 dB - against the reference's fp32 run and against its own fp16 (CPU) run.
    sweep_ramp.json: the same with qp tables that span a 16:1 range (weights.make_state_dict(q_ramp=True)): the reference's
    rate goes 0.38 -> 0.55 -> 1.48 bpp over qp 0 / 32 / 63 (the independent draws of sweep.json give a flat 0.45)."""
    import hashlib
    import sys
    sys.path.insert(0, golden_dir)
    from make_golden_sweep import write_yuv420
    gold = json.load(open(os.path.join(golden_dir, gold_name)))
    cfg = gold["config"]
    q_ramp = bool(cfg.get("q_ramp", False))
    W, H, N = cfg["width"], cfg["height"], cfg["frames"]
    src = str(tmp_path / "seq.yuv")
    write_yuv420(src, W, H, N, cfg["src_seed"])
    assert hashlib.sha256(open(src, "rb").read()).hexdigest() == gold["src_sha256"]
    logs = harness.run_sweep(lambda: _nets(mode, q_ramp), src, W, H, N, qp_i=cfg["qps"], bin_prefix=str(tmp_path / "o"),
                             intra_period=cfg["intra_period"], reset_interval=cfg["reset_interval"], verbose_json=True)
    assert list(logs.keys()) == cfg["qps"]
    if not q_ramp:
        assert cfg["qps"] == harness.sweep_qps(4)
    exact = 0
    for qp in cfg["qps"]:
        got = logs[qp]
        refs = [gold["fp32"][str(qp)]] + ([gold["fp16"][str(qp)]] if mode == "fp16" and "error" not in gold["fp16"] else [])
        assert [k for k in got.keys() if k not in ("qp_i", "qp_p")] == refs[0]["keys"], "log schema differs from the reference's"
        assert got["frame_type"] == refs[0]["log"]["frame_type"] and got["qp_i"] == qp
        blob = open(tmp_path / f"o_q{qp}.bin", "rb").read()
        assert sum(got["frame_bpp"]) * H * W == pytest.approx(8 * len(blob))
        for ref in refs:
            want = ref["log"]
            for fi in range(N):
                gb, wb = got["frame_bpp"][fi] * H * W, want["frame_bpp"][fi] * H * W
                for k in ("frame_psnr", "frame_psnr_y", "frame_psnr_u", "frame_psnr_v"):
                    tol = 1e-4 if mode == "fp32" else 0.05
                    assert abs(got[k][fi] - want[k][fi]) < tol, (qp, fi, k, got[k][fi], want[k][fi])
                if mode == "fp32":
                    assert abs(gb - wb) <= 8, (qp, fi, gb, wb)
                else:       # single ~1.7 KB frames scatter more than the rate point: 4 % per frame, 2 % on the averages below
                    assert abs(gb - wb) <= 0.04 * wb + 8, (qp, fi, gb, wb)
            for k in ("ave_i_frame_bpp", "ave_p_frame_bpp", "ave_all_frame_bpp"):
                rel = 1e-3 if mode == "fp32" else 0.02
                assert got[k] == pytest.approx(want[k], rel=rel), (qp, k)
            for k in ("ave_i_frame_psnr", "ave_p_frame_psnr", "ave_all_frame_psnr", "ave_all_frame_psnr_y"):
                assert abs(got[k] - want[k]) < (1e-4 if mode == "fp32" else 0.05), (qp, k)
        exact += hashlib.sha256(blob).hexdigest() == refs[0]["bin_sha256"]
    if mode == "fp32":
        # whole 10-frame containers byte-identical to the reference's .bin files: a different fp32 summation order can
        # move a value across a rounding boundary and flip one symbol of a frame (same length - checked above), which
        # changes the container's hash; the count is kept with the run's artefacts, one identical container is required
        assert exact >= 1, f"no container byte-identical to the reference's ({exact} of {len(cfg['qps'])})"
    out = os.path.join(os.path.dirname(golden_dir), "..", "gpurun_out")
    os.makedirs(out, exist_ok=True)
    json.dump(dict(points={str(k): v for k, v in logs.items()}, containers_byte_identical=exact),
              open(os.path.join(out, f"sweep{'_ramp' if q_ramp else ''}_{mode}.json"), "w"), indent=1)


@pytest.mark.gpu
def test_manifest_run_two_workers_on_one_gpu_equals_one_worker(tmp_path, golden_dir):
    """BASELINE.json configs[4] on the hardware at hand: the reference's job fan-out (test_video.py:417-532) through
    harness.run_config - a manifest of two sequences x two rate points on a pool of 2 spawned workers sharing GPU 0
    (-w may exceed the GPU count) writes the same .bin files, byte for byte, and the same rate / PSNR numbers as the
    one-worker run, in the reference's log layout."""
    import hashlib
    import sys
    sys.path.insert(0, golden_dir)
    from make_golden_sweep import write_yuv420
    W, H, N = 136, 200, 4
    root = tmp_path / "data"
    (root / "set").mkdir(parents=True)
    for name, seed in (("a_136x200.yuv", 5), ("b_136x200.yuv", 6)):
        write_yuv420(str(root / "set" / name), W, H, N, seed)
    config = {"root_path": str(root), "test_classes": {"S": {"test": 1, "base_path": "set", "src_type": "yuv420", "sequences": {
        "a_136x200.yuv": {"width": W, "height": H, "frames": N, "intra_period": -1},
        "b_136x200.yuv": {"width": W, "height": H, "frames": N, "intra_period": 2}}}}}
    logs, sums = {}, {}
    for w in (1, 2):
        sp = tmp_path / f"bins_w{w}"
        logs[w] = harness.run_config(config, dict(qp_i=[10, 50], stream_path=str(sp), reset_interval=3, record_gpu=True),
                                     workers=w, gpus=1)
        files = sorted(os.listdir(sp / "S"))
        sums[w] = {f: hashlib.sha256(open(os.path.join(sp, "S", f), "rb").read()).hexdigest() for f in files if f.endswith(".bin")}
        # every point's log beside its container, like the reference's worker writes it (test_video.py:345-346,367-368)
        assert [f for f in files if f.endswith(".json")] == [f[:-4] + ".json" for f in sorted(sums[w])]
        point = json.load(open(sp / "S" / "a_136x200.yuv_q10.json"))
        assert point["i_frame_num"] + point["p_frame_num"] == N and "ave_all_frame_bpp" in point
    assert sorted(sums[1]) == ["a_136x200.yuv_q10.bin", "a_136x200.yuv_q50.bin", "b_136x200.yuv_q10.bin", "b_136x200.yuv_q50.bin"]
    assert sums[1] == sums[2]
    for seq in config["test_classes"]["S"]["sequences"]:
        for key in ("000", "001"):
            a, b = logs[1]["S"][seq][key], logs[2]["S"][seq][key]
            assert a["gpu"] == b["gpu"] == 0 and a["qp_i"] == b["qp_i"] == (10, 50)[int(key)]
            for k in ("ave_all_frame_bpp", "ave_all_frame_psnr", "i_frame_num", "p_frame_num"):
                assert a[k] == b[k], (seq, key, k)
    assert logs[1]["S"]["b_136x200.yuv"]["000"]["i_frame_num"] == 2


def test_msssim_matches_reference_values(golden_dir):
    """harness.calc_msssim / calc_msssim_rgb (own restatement of metrics.py:9-79) against values the reference's functions
    produced (tests/golden/make_golden_rgb.py): five scales, four scales (< 176 pixels), the 176 boundary, the RGB mean"""
    g = np.load(os.path.join(golden_dir, "frame_io_rgb.npz"))
    for tag in ("l5", "l4", "edge"):
        assert harness.calc_msssim(g[f"ms_{tag}_a"], g[f"ms_{tag}_b"]) == pytest.approx(float(g[f"ms_{tag}_val"]), abs=1e-12)
    assert harness.calc_msssim_rgb(g["src_c_rgb"], g["rec_c_f32_rgb"]) == pytest.approx(float(g["rec_c_f32_msssim"]), abs=1e-12)
    with pytest.raises(ValueError):
        harness.calc_msssim(np.zeros((64, 200)), np.zeros((64, 200)))        # the reference asserts below 88 pixels


def test_png_sequence_reader(tmp_path, golden_dir):
    import sys
    sys.path.insert(0, golden_dir)
    from make_golden_png import synthetic_rgb, write_png_sequence
    for digits in (1, 5):
        folder = str(tmp_path / f"seq{digits}")
        write_png_sequence(folder, 48, 32, 3, 7, digits=digits)
        r = harness.PNGSequenceReader(folder, 48, 32)
        for fi in range(3):
            (rgb,) = r.read()
            assert rgb.dtype == np.uint8 and rgb.shape == (3, 32, 48) and np.array_equal(rgb, synthetic_rgb(32, 48, fi, 7))
        with pytest.raises(EOFError):
            r.read()
    with pytest.raises(ValueError):
        harness.PNGSequenceReader(str(tmp_path), 48, 32)                      # no im1.png / im00001.png here
    with pytest.raises(ValueError):
        harness.PNGSequenceReader(str(tmp_path / "seq1"), 64, 32).read()      # size mismatch


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["fp32", "fp16"])
def test_png_rate_point_with_msssim_matches_reference(tmp_path, golden_dir, mode):
    """The reference harness's PNG branch (RGB source -> YCbCr around the codec, RGB PSNR, --calc_ssim MS-SSIM): one rate point of
    a 4-frame 96 x 128 sequence against the log the REFERENCE's run_one_point_with_stream wrote for it
    (tests/golden/make_golden_png.py -> png_point.json).  Same keys in the same order; fp32: frames within one byte, PSNR
    within 1e-4 dB, MS-SSIM within 1e-4; fp16: the usual 4 % / 0.05 dB / 5e-3."""
    import sys
    sys.path.insert(0, golden_dir)
    from make_golden_png import write_png_sequence
    gold = json.load(open(os.path.join(golden_dir, "png_point.json")))
    cfg = gold["config"]
    W, H, N = cfg["width"], cfg["height"], cfg["frames"]
    src = str(tmp_path / "seq")
    write_png_sequence(src, W, H, N, cfg["src_seed"])
    i_net, p_net = _nets(mode)
    got = harness.run_one_point(i_net, p_net, src, W, H, N, cfg["qp"], intra_period=cfg["intra_period"],
                                reset_interval=cfg["reset_interval"], bin_path=str(tmp_path / "o.bin"), verbose_json=True,
                                src_type="png", calc_ssim=True, rec_path=str(tmp_path / "rec"))
    refs = [gold["fp32"]] + ([gold["fp16"]] if mode == "fp16" else [])
    assert list(got.keys()) == refs[0]["keys"], "log schema differs from the reference's"
    assert sorted(os.listdir(tmp_path / "rec")) == [f"im{k:05d}.png" for k in range(1, N + 1)]
    for ref in refs:
        want = ref["log"]
        assert got["frame_type"] == want["frame_type"]
        for fi in range(N):
            gb, wb = got["frame_bpp"][fi] * H * W, want["frame_bpp"][fi] * H * W
            assert abs(gb - wb) <= (8 if mode == "fp32" else 0.04 * wb + 8), (fi, gb, wb)
            assert abs(got["frame_psnr"][fi] - want["frame_psnr"][fi]) < (1e-4 if mode == "fp32" else 0.05)
            assert abs(got["frame_msssim"][fi] - want["frame_msssim"][fi]) < (1e-4 if mode == "fp32" else 5e-3)
        for k in ("ave_all_frame_bpp", "ave_all_frame_psnr", "ave_all_frame_msssim"):
            assert got[k] == pytest.approx(want[k], rel=1e-3 if mode == "fp32" else 0.02), k


@pytest.mark.gpu
def test_yuv420_msssim_matches_reference(tmp_path, golden_dir):
    """--calc_ssim on a YUV 4:2:0 source (test_video.py:106-112: MS-SSIM per plane on the clamped reconstruction planes, combined
    (6 Y + U + V) / 8) against the REFERENCE's log of a 176 x 192 sequence (png_point.json, key yuv420_msssim), fp32"""
    import sys
    sys.path.insert(0, golden_dir)
    from make_golden_sweep import write_yuv420
    gold = json.load(open(os.path.join(golden_dir, "png_point.json")))["yuv420_msssim"]
    cfg, want = gold["config"], gold["log"]
    W, H, N = cfg["width"], cfg["height"], cfg["frames"]
    src = str(tmp_path / "seq.yuv")
    write_yuv420(src, W, H, N, cfg["src_seed"])
    i_net, p_net = _nets("fp32")
    got = harness.run_one_point(i_net, p_net, src, W, H, N, cfg["qp"], intra_period=cfg["intra_period"],
                                reset_interval=cfg["reset_interval"], verbose_json=True, calc_ssim=True)
    assert list(got.keys()) == gold["keys"]
    for k in ("frame_msssim", "frame_msssim_y", "frame_msssim_u", "frame_msssim_v", "frame_psnr"):
        for fi in range(N):
            assert abs(got[k][fi] - want[k][fi]) < 1e-4, (k, fi, got[k][fi], want[k][fi])
    assert got["ave_all_frame_msssim"] == pytest.approx(want["ave_all_frame_msssim"], abs=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["fp32", "fp16"])
def test_qp_sweep_at_1080p_matches_reference_rd_points(tmp_path, golden_dir, mode):
    """BASELINE.json configs[2] at its real size: the qp sweep {0, 21, 42, 63} of a 1920 x 1080 YUV 4:2:0 file (4 frames, qp tables
    that span a 16:1 range) through harness.run_sweep against the log the REFERENCE's run_one_point_with_stream wrote for it
    (tests/golden/make_golden_sweep.py --1080p -> sweep_1080p.json; "fp32": its fp32 run, "fp16": its .half() run on the CPU).
    Same log keys; per frame bytes within 0.1 % (fp32) / 0.5 % (fp16) and PSNR (all, y, u, v) within 1e-3 / 0.005 dB; per point
    bpp within 0.05 % / 0.3 %.  The points go to gpurun_out/sweep_1080p_<mode>.json."""
    import hashlib
    import sys
    sys.path.insert(0, golden_dir)
    from make_golden_sweep import write_yuv420
    gold = json.load(open(os.path.join(golden_dir, "sweep_1080p.json")))
    cfg = gold["config"]
    W, H, N = cfg["width"], cfg["height"], cfg["frames"]
    src = str(tmp_path / "seq.yuv")
    write_yuv420(src, W, H, N, cfg["src_seed"])
    assert hashlib.sha256(open(src, "rb").read()).hexdigest() == gold["src_sha256"]
    logs = harness.run_sweep(lambda: _nets(mode, True), src, W, H, N, qp_i=cfg["qps"], intra_period=cfg["intra_period"],
                             reset_interval=cfg["reset_interval"], verbose_json=True)
    tol_b, tol_p, tol_pt = (1e-3, 1e-3, 5e-4) if mode == "fp32" else (5e-3, 5e-3, 3e-3)
    summary = {}
    for qp in cfg["qps"]:
        got, ref = logs[qp], gold[mode][str(qp)]
        want = ref["log"]
        assert [k for k in got.keys() if k not in ("qp_i", "qp_p")] == ref["keys"], "log schema differs from the reference's"
        assert got["frame_type"] == want["frame_type"]
        for fi in range(N):
            assert abs(got["frame_bpp"][fi] / want["frame_bpp"][fi] - 1) <= tol_b, (qp, fi, got["frame_bpp"][fi], want["frame_bpp"][fi])
            for k in ("frame_psnr", "frame_psnr_y", "frame_psnr_u", "frame_psnr_v"):
                assert abs(got[k][fi] - want[k][fi]) < tol_p, (qp, fi, k, got[k][fi], want[k][fi])
        assert abs(got["ave_all_frame_bpp"] / want["ave_all_frame_bpp"] - 1) <= tol_pt, (qp, got["ave_all_frame_bpp"], want["ave_all_frame_bpp"])
        summary[str(qp)] = dict(bpp=got["ave_all_frame_bpp"], bpp_ref=want["ave_all_frame_bpp"], psnr=got["ave_all_frame_psnr"],
                                psnr_ref=want["ave_all_frame_psnr"])
    out = os.path.join(os.path.dirname(golden_dir), "..", "gpurun_out")
    os.makedirs(out, exist_ok=True)
    json.dump(summary, open(os.path.join(out, f"sweep_1080p_{mode}.json"), "w"), indent=1)
