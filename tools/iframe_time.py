"""Developer tool: wall time of the I-frame codec (DMCI) at 1080p, encode and decode, synchronised."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
(ie, pe), (idec, pdec) = bench.load_models(torch.float16, dev, 1, 0)
for m in (ie, idec):
    m.set_use_two_entropy_coders(True)
frames = bench.make_frames(0, torch.float16, dev)[1][:6]
sps = dict(height=1080, width=1920, ec_part=1, use_ada_i=0)
for it in range(6):
    x = frames[it]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    enc = ie.compress(x, 32); torch.cuda.synchronize(); t1 = time.perf_counter()
    dec = idec.decompress(enc["bit_stream"], sps, 32); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"I frame: encode {1e3*(t1-t0):.2f} ms  decode {1e3*(t2-t1):.2f} ms  bytes {len(enc['bit_stream'])}")
