"""Developer tool: what does an I frame cost the two-stage pipeline?  Codes 3 GOPs through EncodeDecodePipeline and prints, around
each I frame, when the encoder emitted each packet and when the decoder handed each picture over (ms relative to the I
packet), plus the mean P-frame interval of each stage."""
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
enc = SequenceEncoder(ie, pe, 32, intra_period=32, reset_interval=32)
dec = SequenceDecoder(idec, pdec, 1080, 1920, True, defer_output=True)
pipe = EncodeDecodePipeline(enc, dec, dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
for rep in range(2):
    tp, tf, kinds = [], [], []
    pipe.run((frames[k % 32] for k in range(n)), lambda p: (tp.append(time.perf_counter()), kinds.append(p.is_i)),
             lambda x: tf.append(time.perf_counter()))
    torch.cuda.synchronize()
tp, tf = np.asarray(tp), np.asarray(tf)
dp, df = np.diff(tp) * 1e3, np.diff(tf) * 1e3
isI = np.asarray(kinds)
print("encoder: median P interval %.3f ms; decoder: median interval %.3f ms; %d frames in %.1f ms = %.1f fps"
      % (np.median(dp), np.median(df), n, (tf[-1] - tp[0]) * 1e3, n / (tf[-1] - tp[0])))
for i in np.nonzero(isI)[0]:
    if i == 0 or i + 8 >= n:
        continue
    t0 = tp[i]
    print("I frame %d: packets  " % i + " ".join("%d:%+.2f" % (k, (tp[k] - t0) * 1e3) for k in range(i - 4, i + 8)))
    print("            pictures " + " ".join("%d:%+.2f" % (k, (tf[k] - t0) * 1e3) for k in range(i - 6, i + 8)))
    # extra wall time charged to the I frame on each stage: the interval sums around it minus the P median
    print("            encoder extra %.2f ms, decoder extra %.2f ms" % (
        (tp[i + 6] - tp[i - 4]) * 1e3 - 10 * np.median(dp), (tf[i + 7] - tf[i - 4]) * 1e3 - 11 * np.median(df)))
