"""Golden vectors for the PNG / RGB path of the reference harness, generated in the BUILD
container by calling the reference's own functions:
  source side   test_video.py:59-63 np_image_to_tensor, src/utils/transforms.py:27-38 rgb2ycbcr, test_video.py:90 the fp16 cast,
                src/layers/cuda_inference.py:174-179 replicate_pad
  decoder side  transforms.py:41-53 ycbcr2rgb, test_video.py:118-126 clamp * 255, calc_psnr, calc_msssim_rgb
  metrics       src/utils/metrics.py:39-79 calc_msssim / calc_msssim_rgb on random planes (5 and 4 levels)
Output: tests/golden/frame_io_rgb.npz (inputs + expected outputs, data only).

    python tests/golden/make_golden_rgb.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def main():
    import torch
    import ref_harness
    ref_harness.load()
    import test_video as tv
    from src.layers.cuda_inference import replicate_pad
    from src.utils.metrics import calc_msssim, calc_msssim_rgb, calc_psnr
    from src.utils.transforms import rgb2ycbcr, ycbcr2rgb
    rng = np.random.default_rng(2025)
    out = {}
    for tag, (h, w) in (("a", (36, 50)), ("b", (16, 32)), ("c", (96, 128))):
        rgb = rng.integers(0, 256, (3, h, w), dtype=np.uint8)
        rgb[:, :2, :4] = np.array([[0, 255, 1, 254]], np.uint8)                    # extremes
        pb, pr = (-h) % 16, (-w) % 16
        x = rgb2ycbcr(tv.np_image_to_tensor(rgb, "cpu"))                           # fp32 [1,3,h,w]
        out[f"src_{tag}_rgb"] = rgb
        out[f"src_{tag}_f32"] = replicate_pad(x, pb, pr).numpy()
        out[f"src_{tag}_f16"] = replicate_pad(x.to(torch.float16), pb, pr).numpy()
        hp, wp = h + pb, w + pr
        for dt, name in ((torch.float32, "f32"), (torch.float16, "f16")):
            if tag == "c":      # a reconstruction that resembles its source (MS-SSIM of unrelated pictures is not a number)
                xh = (replicate_pad(x, pb, pr) + torch.from_numpy(rng.normal(0, 0.03, (1, 3, hp, wp)).astype(np.float32))).to(dt)
            else:               # values outside [0, 1] too
                xh = torch.from_numpy(rng.uniform(-0.1, 1.1, (1, 3, hp, wp)).astype(np.float32)).to(dt)
            rec = torch.clamp(ycbcr2rgb(xh[:, :, :h, :w]) * 255, 0, 255).squeeze(0).cpu().numpy()
            out[f"rec_{tag}_{name}_x"] = xh.numpy()
            out[f"rec_{tag}_{name}_rgb"] = rec
            out[f"rec_{tag}_{name}_psnr"] = np.float64(calc_psnr(rgb, rec))
            if h >= 88 and w >= 88:
                out[f"rec_{tag}_{name}_msssim"] = np.float64(calc_msssim_rgb(rgb, rec))
    # MS-SSIM on single planes: five levels (>= 176) and four (< 176), correlated pairs so that the values are not ~0
    for tag, (h, w) in (("l5", (180, 200)), ("l4", (96, 130)), ("edge", (176, 176))):
        a = rng.integers(0, 256, (h, w)).astype(np.float64)
        a = np.clip(0.5 * a + 0.5 * np.roll(a, 1, 0), 0, 255)
        b = np.clip(a + rng.normal(0, 6.0, (h, w)), 0, 255)
        out[f"ms_{tag}_a"], out[f"ms_{tag}_b"] = a.astype(np.float32), b.astype(np.float32)
        out[f"ms_{tag}_val"] = np.float64(calc_msssim(a.astype(np.float32), b.astype(np.float32)))
    np.savez_compressed(os.path.join(HERE, "frame_io_rgb.npz"), **out)
    print("wrote frame_io_rgb.npz:", len(out), "arrays", os.path.getsize(os.path.join(HERE, "frame_io_rgb.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
