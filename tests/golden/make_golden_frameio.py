"""Golden vectors for the fused frame I/O kernels, generated in the BUILD container by calling the
reference's own functions:
  source side   src/utils/transforms.py:13-24 ycbcr420_to_444_np (scipy zoom, order 0), test_video.py:59-63
                np_image_to_tensor, test_video.py:90 the fp16 cast, src/layers/cuda_inference.py:174-179 replicate_pad
  decoder side  src/utils/transforms.py:56-63 yuv_444_to_420, test_video.py:307-311 clamp * 255, Y rounded / chroma
                truncated to uint8
Output: tests/golden/frame_io.npz (inputs + expected outputs, data only).

    python tests/golden/make_golden_frameio.py
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
    from src.utils.transforms import ycbcr420_to_444_np, yuv_444_to_420
    rng = np.random.default_rng(2024)
    out = {}
    for tag, (h, w) in (("a", (36, 50)), ("b", (16, 32)), ("c", (70, 98))):
        y = rng.integers(0, 256, (1, h, w), dtype=np.uint8)
        uv = rng.integers(0, 256, (2, h // 2, w // 2), dtype=np.uint8)
        pb, pr = (-h) % 16, (-w) % 16
        x = tv.np_image_to_tensor(ycbcr420_to_444_np(y, uv), "cpu")              # fp32 [1,3,h,w]
        out[f"src_{tag}_y"], out[f"src_{tag}_u"], out[f"src_{tag}_v"] = y[0], uv[0], uv[1]
        out[f"src_{tag}_f32"] = replicate_pad(x, pb, pr).numpy()
        out[f"src_{tag}_f16"] = replicate_pad(x.to(torch.float16), pb, pr).numpy()          # test_video.py:90 then :179
        # decoder side: a reconstruction with values outside [0, 1] and exact .5 cases
        hp, wp = h + pb, w + pr
        for dt, name in ((torch.float32, "f32"), (torch.float16, "f16")):
            xh = torch.from_numpy(rng.uniform(-0.1, 1.1, (1, 3, hp, wp)).astype(np.float32))
            xh[0, 0, :2, :8] = torch.tensor([0.5, 1.5, 2.5, 3.5, 126.5, 127.5, 254.5, 255.5]) / 255.0
            xh = xh.to(dt)
            crop = xh[:, :, :h, :w]
            y_rec, uv_rec = yuv_444_to_420(crop)       # (fp16: the reference's tensors are fp16, pooled in fp16)
            y8 = torch.clamp(y_rec * 255, 0, 255).round().to(dtype=torch.uint8).squeeze(0).numpy()
            uv8 = torch.clamp(uv_rec * 255, 0, 255).to(dtype=torch.uint8).squeeze(0).numpy()
            out[f"rec_{tag}_{name}_x"] = xh.numpy()
            out[f"rec_{tag}_{name}_y"], out[f"rec_{tag}_{name}_u"], out[f"rec_{tag}_{name}_v"] = y8[0], uv8[0], uv8[1]
    np.savez_compressed(os.path.join(HERE, "frame_io.npz"), **out)
    print("wrote frame_io.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
