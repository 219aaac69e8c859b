"""Developer tool: cProfile of the sequential decoder loop (1 I + 31 P frames, deferred output) - where the host time of a
decoded frame goes."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from opendcvc_amd.pipeline import SequenceDecoder, SequenceEncoder
torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
(ie, pe), (idec, pdec) = bench.load_models(torch.float16, dev, 1, 0)
for m in (ie, pe, idec, pdec):
    m.set_use_two_entropy_coders(True)
frames = bench.make_frames(0, torch.float16, dev)[1][:32]
enc = SequenceEncoder(ie, pe, 32, intra_period=64, defer_stream=True)
pkts = []
for x in frames:
    pkts += enc.encode(x)
pkts += enc.flush()
for rep in range(2):
    dec = SequenceDecoder(idec, pdec, 1080, 1920, True)
    dec.defer = True
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    if rep == 1:
        pr.enable()
    for p in pkts:
        dec.decode(p)
    dec.flush()
    torch.cuda.synchronize()
    if rep == 1:
        pr.disable()
    print(f"pass {rep}: {1e3 * (time.perf_counter() - t0) / len(pkts):.3f} ms per frame", flush=True)
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
