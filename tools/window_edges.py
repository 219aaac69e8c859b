"""Developer tool: where does a SHORT timed window (the driver's --steps 20) lose time against the steady state?  Codes the
bench's own window (warm-up + alignment, sync, 20 frames through the two-stage pipeline, sync) three times and prints, relative to
the window's start: when every packet left the encoder stage and every picture left the decoder stage, and the end."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from opendcvc_amd.pipeline import EncodeDecodePipeline, SequenceDecoder, SequenceEncoder

torch.set_grad_enabled(False)
torch.set_num_threads(1)
dev = torch.device("cuda", 0)
(ie, pe), (idec, pdec) = bench.load_models(torch.float16, dev, 1, 0)
for m in (ie, pe, idec, pdec):
    m.set_use_two_entropy_coders(True)
frames = bench.make_frames(0, torch.float16, dev)[1]
GOP = bench.GOP
enc = SequenceEncoder(ie, pe, 32, intra_period=GOP, reset_interval=GOP)
dec = SequenceDecoder(idec, pdec, 1080, 1920, True, defer_output=True)
pipe = EncodeDecodePipeline(enc, dec, dev)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
state = {"i": 0}


def run(n, tp=None, tf=None, kinds=None):
    first = state["i"]
    state["i"] += n
    pipe.run((frames[k % GOP] for k in range(first, first + n)),
             (lambda p: (tp.append(time.perf_counter()), kinds.append(p.is_i))) if tp is not None else None,
             (lambda x: tf.append(time.perf_counter())) if tf is not None else None)


run(GOP + bench.window_start(K))           # captures, warm-up, alignment: the next frame is where the bench starts its window
for rep in range(3):
    tp, tf, kinds = [], [], []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(K, tp, tf, kinds)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    tp, tf = (np.asarray(tp) - t0) * 1e3, (np.asarray(tf) - t0) * 1e3
    print("window %d: %d frames in %.2f ms = %.1f fps" % (rep, K, (t1 - t0) * 1e3, K / (t1 - t0)))
    print("  packets  " + " ".join(("I" if k else "") + "%.2f" % t for t, k in zip(tp, kinds)))
    print("  pictures " + " ".join("%.2f" % t for t in tf))
    print("  packet intervals  " + " ".join("%.2f" % t for t in np.diff(tp)))
    print("  picture intervals " + " ".join("%.2f" % t for t in np.diff(tf)))
    run(GOP - K)                              # back to the same place in the GOP
