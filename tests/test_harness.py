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
