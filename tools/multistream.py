"""Developer experiment: S independent 1080p streams on ONE GPU, each the bench's two-stage encode/decode pipeline
(own models, own host threads and HIP streams).  If the aggregate rate grows with S the single-stream pipeline leaves
the GPU idle somewhere; if it stays flat the GPU is saturated by one stream.

    python3 tools/multistream.py 1 2 3
"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from opendcvc_amd.pipeline import EncodeDecodePipeline, SequenceDecoder, SequenceEncoder

torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
GOP, QP = bench.GOP, bench.QP


def make_stream():
    (ie, pe), (idec, pdec) = bench.load_models(torch.float16, dev, 1, 0)
    for m in (ie, pe, idec, pdec):
        m.set_use_two_entropy_coders(True)
    enc = SequenceEncoder(ie, pe, QP, intra_period=GOP, reset_interval=GOP)
    dec = SequenceDecoder(idec, pdec, bench.HEIGHT, bench.WIDTH, True, defer_output=True)
    return EncodeDecodePipeline(enc, dec, dev)


frames = bench.make_frames(0, torch.float16, dev)[1]
for S in [int(a) for a in sys.argv[1:]] or [1, 2]:
    pipes = [make_stream() for _ in range(S)]

    def run(n):
        ts = [threading.Thread(target=p.run, args=((frames[k % GOP] for k in range(n)),)) for p in pipes]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        torch.cuda.synchronize()

    run(GOP)            # captures + warm-up (one whole GOP, so the encoder's frame counter is back on an I frame)
    n = 4 * GOP
    t0 = time.perf_counter()
    run(n)
    dt = time.perf_counter() - t0
    print(f"{S} stream(s) per GPU: {S * n / dt:.1f} frames/s aggregate ({n / dt:.1f} per stream)", flush=True)
    del pipes
    torch.cuda.empty_cache()
