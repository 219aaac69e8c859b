"""Developer tool: N frames encode -> sync -> decode -> sync (for rocprofv3 --kernel-trace gap analysis)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from opendcvc_amd.pipeline import SequenceDecoder, SequenceEncoder
torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
(ie, pe), (idec, pdec) = bench.load_models(torch.float16, dev, 1, 0)
for m in (ie, pe, idec, pdec):
    m.set_use_two_entropy_coders(True)
frames = bench.make_frames(0, torch.float16, dev)[:12]
enc = SequenceEncoder(ie, pe, 32, intra_period=32, reset_interval=32)
dec = SequenceDecoder(idec, pdec, 1080, 1920, True)
for x in frames:
    p = enc.encode(x); torch.cuda.synchronize()
    dec.decode(p); torch.cuda.synchronize()
