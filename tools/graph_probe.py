"""Developer probe: does a run of DepthConvBlocks captured in a HIP graph (torch.cuda.CUDAGraph) replay
correctly, and what do eager launches vs one graph launch cost on host and GPU?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
(ie, pe), _ = bench.load_models(torch.float16, dev, 1, 0)
pe._ensure_layers()
x = (torch.randn((136, 240, 256), device=dev) * 0.5).half()

def run(xin):
    return pe._extractor_part2(xin)

s = torch.cuda.Stream(dev)
with torch.cuda.stream(s):
    for _ in range(3):
        ref = run(x)
    s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        out = run(x)
    s.synchronize()
    g.replay(); s.synchronize()
    print("graph == eager:", torch.equal(out, ref))
    for name, fn in (("eager", lambda: run(x)), ("graph", g.replay)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.synchronize()
        t0 = time.perf_counter(); e0.record()
        for _ in range(50):
            fn()
        e1.record(); t1 = time.perf_counter()
        s.synchronize()
        print(f"{name}: host {1e3*(t1-t0)/50:.3f} ms/iter  gpu {e0.elapsed_time(e1)/50:.3f} ms/iter")
