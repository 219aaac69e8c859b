"""Fused frame I/O kernels against the reference's I/O semantics restated in numpy/torch-CPU
(src/utils/transforms.py:13-24,56-63; test_video.py:60-63,90,179,307-311)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from opendcvc_amd.pipeline import load_yuv420_frame, store_yuv420_frame

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("h,w", [(1080, 1920), (36, 50)])
def test_yuv420_to_padded_frame(h, w, dtype):
    rng = np.random.default_rng(1)
    y = rng.integers(0, 256, (h, w), dtype=np.uint8)
    u = rng.integers(0, 256, (h // 2, w // 2), dtype=np.uint8)
    v = rng.integers(0, 256, (h // 2, w // 2), dtype=np.uint8)
    got = load_yuv420_frame(torch.from_numpy(y).cuda(), torch.from_numpy(u).cuda(), torch.from_numpy(v).cuda(), dtype)
    up = lambda a: np.repeat(np.repeat(a, 2, 0), 2, 1)                 # scipy zoom order 0 by exactly 2
    x = np.stack([y, up(u), up(v)]).astype(np.float32)[None]
    ref = (torch.from_numpy(x) / 255.0).to(dtype)                        # np_image_to_tensor + x.to(float16)
    pb, pr = (-h) % 16, (-w) % 16
    ref = F.pad(ref.float(), (0, pr, 0, pb), mode="replicate").to(dtype)
    assert torch.equal(got.cpu(), ref)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_frame_to_yuv420(dtype):
    rng = np.random.default_rng(2)
    h, w, hp, wp = 36, 50, 48, 64
    x = torch.from_numpy(rng.uniform(-0.1, 1.1, (1, 3, hp, wp)).astype(np.float32)).to(dtype)
    yy, uu, vv = store_yuv420_frame(x.cuda(), h, w)
    xc = x[:, :, :h, :w]
    y_rec = torch.clamp(xc[:, :1] * 255, 0, 255).round().to(torch.uint8)[0, 0]
    uv = F.avg_pool2d(xc[:, 1:].float(), 2, 2).to(dtype)
    uv_rec = torch.clamp(uv * 255, 0, 255).to(torch.uint8)[0]
    assert torch.equal(yy.cpu(), y_rec) and torch.equal(uu.cpu(), uv_rec[0]) and torch.equal(vv.cpu(), uv_rec[1])
