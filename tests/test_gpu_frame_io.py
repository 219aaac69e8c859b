"""Fused frame I/O kernels against fixtures produced by the reference's own functions
(tests/golden/make_golden_frameio.py: src/utils/transforms.py:13-24,56-63, test_video.py:59-63,90,179,307-311),
plus the full-size property the fixtures are too small for."""
import os

import numpy as np
import pytest
import torch

from opendcvc_amd.pipeline import load_yuv420_frame, store_yuv420_frame

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "frame_io.npz"))


@pytest.mark.parametrize("dtype,name", [(torch.float32, "f32"), (torch.float16, "f16")])
@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_yuv420_to_padded_frame_matches_reference(gold, tag, dtype, name):
    y, u, v = (torch.from_numpy(gold[f"src_{tag}_{k}"]).cuda() for k in "yuv")
    got = load_yuv420_frame(y, u, v, dtype)
    want = gold[f"src_{tag}_{name}"]
    assert got.dtype == dtype and tuple(got.shape) == want.shape
    assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize("dtype,name", [(torch.float32, "f32"), (torch.float16, "f16")])
@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_frame_to_yuv420_matches_reference(gold, tag, dtype, name):
    x = torch.from_numpy(gold[f"rec_{tag}_{name}_x"]).cuda()
    h, w = gold[f"rec_{tag}_{name}_y"].shape
    yy, uu, vv = store_yuv420_frame(x, h, w)
    for got, k in ((yy, "y"), (uu, "u"), (vv, "v")):
        assert np.array_equal(got.cpu().numpy(), gold[f"rec_{tag}_{name}_{k}"]), k


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_full_size_round_trip_property(dtype):
    """1080p: planes -> padded frame -> planes gives the source back (Y exactly; chroma exactly, being constant over
    each 2x2 block), and the padding rows / columns replicate the last picture row / column."""
    h, w = 1080, 1920
    rng = np.random.default_rng(1)
    y = torch.from_numpy(rng.integers(0, 256, (h, w), dtype=np.uint8)).cuda()
    u = torch.from_numpy(rng.integers(0, 256, (h // 2, w // 2), dtype=np.uint8)).cuda()
    v = torch.from_numpy(rng.integers(0, 256, (h // 2, w // 2), dtype=np.uint8)).cuda()
    x = load_yuv420_frame(y, u, v, dtype)
    assert tuple(x.shape) == (1, 3, 1088, 1920)
    assert torch.equal(x[:, :, h:, :], x[:, :, h - 1:h, :].expand(-1, -1, 8, -1))
    y2, u2, v2 = store_yuv420_frame(x, h, w, round_uv=True)
    assert torch.equal(y2, y) and torch.equal(u2, u) and torch.equal(v2, v)


# ------------------------------------------------------------------------------------------- RGB (PNG sources)
@pytest.fixture(scope="module")
def gold_rgb(golden_dir):
    return np.load(os.path.join(golden_dir, "frame_io_rgb.npz"))


@pytest.mark.parametrize("dtype,name", [(torch.float32, "f32"), (torch.float16, "f16")])
@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_rgb_to_padded_frame_matches_reference(gold_rgb, tag, dtype, name):
    """uint8 RGB -> /255 -> rgb2ycbcr (BT.709, fp32, the reference's operation order) -> clamp -> cast -> replicate pad, bit for
    bit what np_image_to_tensor + rgb2ycbcr + .to(float16) + replicate_pad give (tests/golden/make_golden_rgb.py)"""
    from opendcvc_amd.harness import load_rgb_frame
    got = load_rgb_frame(torch.from_numpy(gold_rgb[f"src_{tag}_rgb"]).cuda(), dtype)
    want = gold_rgb[f"src_{tag}_{name}"]
    assert got.dtype == dtype and tuple(got.shape) == want.shape
    assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize("dtype,name", [(torch.float32, "f32"), (torch.float16, "f16")])
@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_frame_to_rgb_and_its_metrics_match_reference(gold_rgb, tag, dtype, name):
    """clamp(ycbcr2rgb(x_hat) * 255, 0, 255) in the reconstruction's own dtype (fp16: every operation rounded like the reference's
    half tensors), then the reference's RGB PSNR and MS-SSIM of it"""
    from opendcvc_amd.harness import reconstruct_rgb, rgb_distortion
    x = torch.from_numpy(gold_rgb[f"rec_{tag}_{name}_x"]).cuda()
    rgb = torch.from_numpy(gold_rgb[f"src_{tag}_rgb"]).cuda()
    _, h, w = rgb.shape
    rec = reconstruct_rgb(x, h, w)
    want = gold_rgb[f"rec_{tag}_{name}_rgb"]
    assert rec.dtype == dtype and np.array_equal(rec.cpu().numpy(), want)
    with_ssim = f"rec_{tag}_{name}_msssim" in gold_rgb.files
    psnr, ms = rgb_distortion(x, rgb, calc_ssim=with_ssim)
    assert psnr[0] == pytest.approx(float(gold_rgb[f"rec_{tag}_{name}_psnr"]), abs=1e-9)
    if with_ssim:
        assert ms[0] == pytest.approx(float(gold_rgb[f"rec_{tag}_{name}_msssim"]), abs=1e-12)
