"""Developer tool: when does the cgroup throttle a bench-like run?  Prints nr_throttled / throttled_usec of /sys/fs/cgroup/cpu.stat
and the number of threads of this process after each phase (model build, pipeline warm-up, timed window, sequential decode)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def stat(tag, t0=[time.perf_counter()]):
    d = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
    print("%-28s t=%6.2fs nr_throttled %s throttled %.2f s usage %.2f s threads %d" % (
        tag, time.perf_counter() - t0[0], d.get("nr_throttled"), int(d.get("throttled_usec", 0)) / 1e6,
        int(d.get("usage_usec", 0)) / 1e6, len(os.listdir("/proc/self/task"))), flush=True)


stat("start")
import numpy as np, torch
stat("import torch")
torch.set_num_threads(1)
import bench
from opendcvc_amd.pipeline import EncodeDecodePipeline, SequenceDecoder, SequenceEncoder
torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
stat("gpu init")
(ie, pe), (idec, pdec) = bench.load_models(torch.float16, dev, 1, 0)
stat("models built")
for m in (ie, pe, idec, pdec):
    m.set_use_two_entropy_coders(True)
frames = bench.make_frames(0, torch.float16, dev)[1]
stat("frames made")
enc = SequenceEncoder(ie, pe, 32, intra_period=32, reset_interval=32)
dec = SequenceDecoder(idec, pdec, 1080, 1920, True, defer_output=True)
pipe = EncodeDecodePipeline(enc, dec, dev)
pipe.run((frames[k % 32] for k in range(40)))
stat("pipeline warm-up (40)")
for rep in range(4):
    t0 = time.perf_counter()
    pipe.run((frames[k % 32] for k in range(24, 24 + 64)))
    torch.cuda.synchronize()
    stat("pipelined 64: %.1f fps" % (64 / (time.perf_counter() - t0)))
