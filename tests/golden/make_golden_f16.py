"""fp16 layer-level fixtures from the REFERENCE ITSELF: runs the reference's DepthConvBlock /
SubpelConv2x / nn.Conv2d in .half() on the CPU (torch fallback path, forward_torch: src/layers/layers.py:44-51,92-106 -
the arithmetic the reference's fp16 GPU path follows: fp16 storage of every intermediate, rounded after every conv and
after the activation) on seeded weights and inputs, and writes tests/golden/ops_small_f16.npz:

    <case>.y16    the reference's fp16 result [1, C, H', W'] (the large-map case: its first 16 channels)

Weights, input and quant step are regenerated from the case's seed on the GPU box (tests/layer_utils.py: f16_case_inputs),
so the fixture carries expected outputs only.  The fp32 value both fp16 implementations approximate is computed by the
test through the fp32 oracle on the same fp16-rounded weights; this script prints how far the reference's own fp16 run is
from the reference's fp32 run for orientation.  Build container only.

    python tests/golden/make_golden_f16.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, REPO)

import ref_harness  # noqa: E402
from layer_utils import F16_CONV_CASES, F16_DCB_CASES, F16_LARGE_KEEP, f16_conv_inputs, f16_dcb_inputs  # noqa: E402


def t2n(t):
    return t.detach().cpu().numpy()


def load(mod, sd, prefix):
    """weights rounded to fp16 first: the fp32 run then sees exactly the values the .half() run computes with"""
    own = {k[len(prefix) + 1:]: torch.from_numpy(v.astype(np.float16).astype(np.float32)) for k, v in sd.items()}
    mod.load_state_dict(own)
    return mod.eval()


def main():
    _, _, L, _, _, _ = ref_harness.load()
    torch.set_grad_enabled(False)
    torch.set_num_threads(8)
    out = {}
    def report(name, y16, y32):
        rms = float(y32.pow(2).mean().sqrt())
        d = (y16.float() - y32).abs()
        print(f"{name:16s} rms {rms:.3f}  reference fp16 vs reference fp32: max {float(d.max()) / rms:.2e} "
              f"mean {float(d.mean()) / rms:.2e} (x rms)")

    for case in F16_DCB_CASES:
        name, cin, c, adaptor, shortcut, quant, H, W, seed = case
        sd, x, q = f16_dcb_inputs(case)
        m = load(L.DepthConvBlock(cin, c, shortcut=shortcut), sd, "m")
        qt = None if q is None else torch.from_numpy(q.reshape(1, c, 1, 1))
        y32 = m.forward_torch(torch.from_numpy(x).float(), quant_step=qt)
        m.half()
        y16 = m.forward_torch(torch.from_numpy(x), quant_step=None if qt is None else qt.half())
        report(name, y16, y32)
        out[name + ".y16"] = t2n(y16)[:, :F16_LARGE_KEEP] if H * W >= 12000 else t2n(y16)
    for case in F16_CONV_CASES:
        name, kind, cin, cout, k, H, W, seed = case
        sd, x = f16_conv_inputs(case)
        if kind == "subpel":
            m = load(L.SubpelConv2x(cin, cout, k, padding=k // 2), sd, "m")
            run = m.forward_torch
        else:
            stride, pad = (2, 1 if k == 3 else 0) if kind == "s2" else (1, 0)
            m = load(torch.nn.Conv2d(cin, cout, k, stride=stride, padding=pad), sd, "m")
            run = m
        y32 = run(torch.from_numpy(x).float())
        m.half()
        y16 = run(torch.from_numpy(x))
        report(name, y16, y32)
        out[name + ".y16"] = t2n(y16)
    path = os.path.join(HERE, "ops_small_f16.npz")
    np.savez_compressed(path, **out)
    print("written", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
