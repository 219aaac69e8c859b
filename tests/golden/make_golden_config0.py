"""BASELINE.json configs[0] to the letter: DCVC-RT-Intra on a single 256 x 256 RGB frame, q = 32, on the reference's PyTorch CPU
fallback path - the REFERENCE's DMCI (fp32, torch fallback ops, its own rANS coder) on an RGB picture prepared and scored with the
reference harness's own functions (np_image_to_tensor, rgb2ycbcr, ycbcr2rgb, calc_psnr; test_video.py:84-90,116-120).
Output: tests/golden/config0_rgb.json (stream bytes / sha256, RGB PSNR, the reconstruction's uint8 sha256; data only).

    python tests/golden/make_golden_config0.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)

import make_golden as G  # noqa: E402
import ref_harness  # noqa: E402
from make_golden_png import synthetic_rgb  # noqa: E402


def main():
    DMC, DMCI, *_ = ref_harness.load()
    import test_video as tv
    torch.set_grad_enabled(False)
    i_net, _ = G.load_models(DMC, DMCI)
    i_net.set_use_two_entropy_coders(False)
    rgb = synthetic_rgb(256, 256, 0, 41)
    x = tv.rgb2ycbcr(tv.np_image_to_tensor(rgb, "cpu"))
    enc = i_net.compress(x, 32)
    dec = i_net.decompress(enc["bit_stream"], dict(height=256, width=256, ec_part=0, use_ada_i=0), 32)
    assert torch.equal(dec["x_hat"], enc["x_hat"])
    rec = torch.clamp(tv.ycbcr2rgb(dec["x_hat"]) * 255, 0, 255).squeeze(0).numpy()
    out = dict(size=256, qp=32, src_seed=41, seed=G.SEED, thres=G.THRES, bytes=len(enc["bit_stream"]), sha256=G.sha(enc["bit_stream"]),
               psnr_rgb=float(tv.calc_psnr(rgb, rec)), rec_u8_sha256=G.sha(np.round(rec).astype(np.uint8).tobytes()))
    json.dump(out, open(os.path.join(HERE, "config0_rgb.json"), "w"), indent=1)
    print(out)


if __name__ == "__main__":
    main()
