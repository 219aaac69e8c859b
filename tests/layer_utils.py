"""Seeded layer weights shared by the GPU layer tests and the fp16 reference-fixture generator
(tests/golden/make_golden_f16.py): fixtures then only carry inputs and expected outputs - the weights are regenerated
from (seed, shape) on the GPU box."""
import numpy as np


def make_dcb_weights(rng, prefix, cin, c, adaptor):
    """state-dict entries of one DepthConvBlock (reference keys: layers.py:65-90), O(1) activations"""
    sd = {}

    def conv(name, co, ci, k=1, gain=1.0):
        sd[f"{prefix}.{name}.weight"] = (rng.standard_normal((co, ci, k, k)) * gain / np.sqrt(ci * k * k)).astype(np.float32)
        sd[f"{prefix}.{name}.bias"] = (rng.standard_normal(co) * 0.1).astype(np.float32)

    if adaptor:
        conv("adaptor", c, cin)
    conv("dc.0", c, c)
    sd[f"{prefix}.dc.2.weight"] = (rng.standard_normal((c, 1, 3, 3)) / 3).astype(np.float32)
    sd[f"{prefix}.dc.2.bias"] = (rng.standard_normal(c) * 0.1).astype(np.float32)
    conv("dc.3", c, c, gain=0.5)
    conv("ffn.0", 4 * c, c)
    conv("ffn.2", c, 2 * c, gain=0.5)
    return sd


def make_conv_weights(rng, prefix, cout, cin, k):
    return {f"{prefix}.weight": (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32),
            f"{prefix}.bias": (rng.standard_normal(cout) * 0.1).astype(np.float32)}


# fp16 layer cases pinned against the REFERENCE's own .half() run (tests/golden/ops_small_f16.npz):
# name, cin, c, adaptor, shortcut, quant, H, W, seed
F16_DCB_CASES = [
    ("dcb256", 256, 256, False, False, False, 16, 24, 9001),
    ("dcb256_q", 256, 256, False, False, True, 13, 21, 9002),
    ("dcb256_ad512", 512, 256, True, False, False, 9, 17, 9003),
    ("dcb128_short", 128, 128, False, True, False, 5, 7, 9004),
    ("dcb368_q", 368, 368, False, False, True, 10, 11, 9005),
    ("dcb320_ad256", 256, 320, True, False, False, 8, 10, 9006),
    ("dcb384_ad512", 512, 384, True, False, False, 7, 8, 9007),
    ("dcb512", 512, 512, False, False, False, 5, 9, 9008),
    ("dcb256_large", 256, 256, False, True, True, 96, 130, 9009),       # 12 480 px: the 128-pixel-tile kernels
]
# name, kind, cin, cout (after the shuffle for subpel), k, H, W, seed
F16_CONV_CASES = [
    ("subpel1", "subpel", 128, 128, 1, 9, 15, 9101),
    ("subpel3", "subpel", 128, 256, 3, 8, 12, 9102),
    ("conv3s2", "s2", 256, 128, 3, 14, 22, 9103),
    ("conv2s2", "s2", 128, 128, 2, 12, 20, 9104),
    ("conv1x1", "plain", 384, 256, 1, 9, 13, 9105),
]
F16_LARGE_KEEP = 16        # channels of the large-map case kept in the fixture


def f16_dcb_inputs(case):
    """(state dict, x fp16 [1, cin, H, W], q fp32 [c] or None) of a F16_DCB_CASES entry - the same draw order in the
    generator (build container) and in the GPU test"""
    name, cin, c, adaptor, shortcut, quant, H, W, seed = case
    rng = np.random.default_rng(seed)
    sd = make_dcb_weights(rng, "m", cin, c, adaptor)
    x = rng.standard_normal((1, cin, H, W)).astype(np.float16)
    q = rng.uniform(0.5, 1.5, c).astype(np.float32) if quant else None
    return sd, x, q


def f16_conv_inputs(case):
    name, kind, cin, cout, k, H, W, seed = case
    rng = np.random.default_rng(seed)
    prefix = "m.conv.0" if kind == "subpel" else "m"
    sd = make_conv_weights(rng, prefix, cout * 4 if kind == "subpel" else cout, cin, k)
    x = rng.standard_normal((1, cin, H, W)).astype(np.float16)
    return sd, x
