"""Developer micro-benchmark: HIP-event time of the two DepthConvBlock kernels at a given shape."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from opendcvc_amd import _lib, nn as L


def make_dcb_weights(rng, prefix, cin, c, adaptor):
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

def run(C, H, W, dtype=torch.float16, iters=30):
    rng = np.random.default_rng(0)
    sd = make_dcb_weights(rng, "m", C, C, False)
    blk = L.DepthConvBlock(sd, "m", dtype)
    x = (torch.randn((H, W, blk.c_p), device="cuda") * 0.5).to(dtype)
    out = torch.empty_like(x)
    lib = _lib.lib()
    scratch = L.Scratch.get(lib.dcvc_dcb_scratch_bytes(blk.h, H, W), x.device)
    head, tail = ctypes.c_float(), ctypes.c_float()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for it in (5, iters):
        _lib.check(lib.dcvc_dcb_profile(blk.h, L._p(x), blk.c_p, blk.c_p, H, W, L._p(out), blk.c_p, L._p(scratch), st, it, ctypes.byref(head), ctypes.byref(tail)))
    flop = 2.0 * H * W * (7 * C * C + 9 * C)
    print(f"C={C} {H}x{W} {dtype}: head {head.value*1e3:.1f} us  tail {tail.value*1e3:.1f} us  tail {flop/tail.value/1e9:.1f} TFLOP/s  ablate={os.environ.get('DCVC_ABLATE','0')}", flush=True)

if __name__ == "__main__":
    shapes = [(256, 136, 240), (256, 68, 120), (384, 68, 120), (128, 17, 30), (320, 136, 240)]
    for C, H, W in shapes if len(sys.argv) < 2 else [tuple(int(v) for v in sys.argv[1:4])]:
        run(C, H, W)
