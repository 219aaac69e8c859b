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

def run_cold(C, H, W, nblk=8, dtype=torch.float16, iters=10):
    """tails of nblk different blocks in turn (weights of one block no longer sit in L2 when it runs again), `a` precomputed"""
    rng = np.random.default_rng(0)
    blks = [L.DepthConvBlock(make_dcb_weights(rng, "m", C, C, False), "m", dtype) for _ in range(nblk)]
    x = (torch.randn((H, W, blks[0].c_p), device="cuda") * 0.5).to(dtype)
    out = torch.empty_like(x)
    lib = _lib.lib()
    scratch = L.Scratch.get(lib.dcvc_dcb_scratch_bytes(blks[0].h, H, W), x.device)
    head, tail = ctypes.c_float(), ctypes.c_float()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    tot = 0.0
    for rep in range(iters + 2):
        for b in blks:
            _lib.check(lib.dcvc_dcb_profile(b.h, L._p(x), b.c_p, b.c_p, H, W, L._p(out), b.c_p, L._p(scratch), st, 1, ctypes.byref(head), ctypes.byref(tail)))
            if rep >= 2:
                tot += tail.value
    print(f"C={C} {H}x{W} {nblk} blocks in turn: tail {tot / (iters * nblk) * 1e3:.1f} us per launch", flush=True)


def run_chain(C, H, W, n=4, dtype=torch.float16, iters=20):
    """a run of n blocks as the codec launches it (every tail but the last carries the next block's head)"""
    rng = np.random.default_rng(0)
    blks = [L.DepthConvBlock(make_dcb_weights(rng, "m", C, C, False), "m", dtype) for _ in range(n)]
    x = (torch.randn((H, W, blks[0].c_p), device="cuda") * 0.5).to(dtype)
    out = torch.empty_like(x)
    for _ in range(3):
        L.dcb_chain(blks, x, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        L.dcb_chain(blks, x, out=out)
    e1.record()
    torch.cuda.synchronize()
    print(f"C={C} {H}x{W} chain of {n}: {e0.elapsed_time(e1) / iters * 1e3:.1f} us (1 head + {n} tails, {n - 1} with a fused head)", flush=True)
    if os.environ.get("KBENCH_GRAPH"):      # the same launches replayed from a captured HIP graph
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.graph(g, stream=side):
            L.dcb_chain(blks, x, out=out)
        torch.cuda.synchronize()
        for _ in range(3):
            g.replay()
        e0.record()
        for _ in range(iters):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        print(f"   ... replayed from a HIP graph: {e0.elapsed_time(e1) / iters * 1e3:.1f} us", flush=True)


def run_mix(iters=40):
    """small-map chains between large-map ones (the GPU's clocks then sit where they sit inside a frame, not where a loop of
    light kernels lets them climb); read the per-kernel durations from a kernel trace"""
    rng = np.random.default_rng(0)
    big = [L.DepthConvBlock(make_dcb_weights(rng, "m", 256, 256, False), "m", torch.float16) for _ in range(4)]
    small = [L.DepthConvBlock(make_dcb_weights(rng, "m", 384, 384, False), "m", torch.float16) for _ in range(4)]
    xb = (torch.randn((136, 240, 256), device="cuda") * 0.5).half()
    xs = (torch.randn((68, 120, 384), device="cuda") * 0.5).half()
    ob, os_ = torch.empty_like(xb), torch.empty_like(xs)
    for _ in range(iters):
        L.dcb_chain(big, xb, out=ob)
        L.dcb_chain(big, xb, out=ob)
        L.dcb_chain(small, xs, out=os_)
    torch.cuda.synchronize()


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "mix":
    run_mix()


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "chain":
    if len(sys.argv) > 5:          # chain C n H W
        run_chain(int(sys.argv[2]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[3]))
    elif len(sys.argv) > 3:
        run_chain(int(sys.argv[2]), 68, 120, int(sys.argv[3]))
    else:
        for C in (256, 384):
            run_chain(C, 68, 120, 4)
            run_chain(C, 68, 120, 1)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "cold":
    for C in (256, 384):
        run_cold(C, 68, 120)
        run_cold(C, 68, 120, nblk=1)
    run_cold(256, 136, 240)
    run_cold(256, 136, 240, nblk=1)


if __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] in ("conv", "adapt", "cold", "chain", "mix")):
    shapes = [(256, 136, 240), (256, 68, 120), (384, 68, 120), (128, 17, 30), (320, 136, 240), (384, 136, 240), (512, 68, 120)]
    for C, H, W in shapes if len(sys.argv) < 2 else [tuple(int(v) for v in sys.argv[1:4])]:
        run(C, H, W)


def run_conv(cin, cout, k, stride, pad, epi, H, W, dtype=torch.float16, iters=30):
    rng = np.random.default_rng(0)
    sd = {"m.weight": (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32),
          "m.bias": (rng.standard_normal(cout) * 0.1).astype(np.float32)}
    conv = L.Conv2d(sd, "m", dtype, stride, pad, epi)
    x = (torch.randn((H, W, conv.cin_p), device="cuda") * 0.5).to(dtype)
    q = torch.ones(cout, device="cuda")
    out = conv(x, quant=q if epi == 1 else None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        conv(x, quant=q if epi == 1 else None, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    Ho, Wo = conv.out_hw(H, W)
    if epi == 2:
        Ho, Wo = Ho // 2, Wo // 2
    flop = 2.0 * Ho * Wo * cin * cout * k * k
    print(f"conv {cin}->{cout} k{k} s{stride} epi{epi} {H}x{W}: {ms*1e3:.1f} us  {flop/ms/1e9:.1f} TFLOP/s", flush=True)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "conv":
    for args in [(128, 1024, 3, 1, 1, 2, 68, 120), (256, 128, 3, 2, 1, 0, 136, 240), (384, 384, 1, 1, 0, 0, 68, 120),
                 (384, 256, 1, 1, 0, 0, 68, 120), (256, 256, 2, 2, 0, 0, 136, 240), (256, 256, 1, 1, 0, 1, 136, 240),
                 (192, 256, 1, 1, 0, 0, 136, 240), (320, 192, 1, 1, 0, 0, 136, 240), (128, 512, 1, 1, 0, 2, 17, 30),
                 (128, 128, 2, 2, 0, 0, 68, 120)]:
        run_conv(*args)


def run_adapt(cin0, cin1, C, H, W, dtype=torch.float16, iters=30):
    """two-source DepthConvBlock with adaptor (e.g. encoder.conv2.0: features | context)"""
    rng = np.random.default_rng(0)
    sd = make_dcb_weights(rng, "m", cin0 + cin1, C, True)
    blk = L.DepthConvBlock(sd, "m", dtype)
    x0 = (torch.randn((H, W, cin0), device="cuda") * 0.5).to(dtype)
    x1 = (torch.randn((H, W, cin1), device="cuda") * 0.5).to(dtype)
    out = blk(x0, x1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        blk(x0, x1, out=out)
    e1.record()
    torch.cuda.synchronize()
    print(f"adaptor block {cin0}+{cin1}->{C} {H}x{W}: {e0.elapsed_time(e1) / iters * 1e3:.1f} us (head + tail)", flush=True)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "adapt":
    run_adapt(256, 256, 256, 136, 240)
    run_adapt(256, 256, 320, 136, 240)
    run_adapt(128, 384, 384, 68, 120)
