"""Developer probe: the real model's y_prior_fusion chain (its own weight buffers) on random input, in a loop - are the
small-map tails as fast here as in tools/kbench.py chain?  Read the per-kernel durations from a kernel trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from opendcvc_amd import nn as L
torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
(ie, pe), (idec, pdec) = bench.load_models(torch.float16, dev, 1, 0)
pe._ensure_layers()
n = pe._layers
mode = sys.argv[1] if len(sys.argv) > 1 else "random"
cat = (torch.randn((68, 120, 384), device=dev) * 0.5).half()
if mode == "zeros":
    cat.zero_()
for _ in range(30):
    out = L.dcb_chain(n["fusion"], cat, then_conv=n["fusion_out"])
torch.cuda.synchronize()
print("fusion blocks:", [(b.cin, b.c, b.has_adaptor) for b in n["fusion"]], "out", tuple(out.shape), float(out.float().abs().mean()))
