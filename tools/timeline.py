"""Developer tool: where does a 1080p P-frame encode / decode spend its wall time?  Wraps the model's
internal steps with host timers (launch time = host time to enqueue, wait = time blocked in syncs)."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from opendcvc_amd import _lib
from opendcvc_amd.pipeline import SequenceDecoder, SequenceEncoder

torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
(ie, pe), (idec, pdec) = bench.load_models(torch.float16, dev, 1, 0)
for m in (ie, pe, idec, pdec):
    m.set_use_two_entropy_coders(True)
frames = bench.make_frames(0, torch.float16, dev)[:12]
enc = SequenceEncoder(ie, pe, 32, intra_period=32, reset_interval=32)
dec = SequenceDecoder(idec, pdec, 1080, 1920, True)

# monkeypatch timers
T = {}
def timed(obj, name, label=None):
    f = getattr(obj, name)
    label = label or name
    def w(*a, **k):
        t0 = time.perf_counter()
        r = f(*a, **k)
        T[label] = T.get(label, 0.0) + time.perf_counter() - t0
        return r
    setattr(obj, name, w)

for m, tag in ((pe, "enc"), (pdec, "dec")):
    ec = m.entropy_coder
    for n in ("encode_y", "encode_z", "flush", "get_encoded_stream", "set_stream", "decode_and_get_y", "get_decoded", "reset"):
        timed(ec, n, f"{tag}.ec.{n}")
    for n in ("_stage_q", "_stage_reference", "_unshuffle8"):
        timed(m, n, f"{tag}.{n}")
    run = m._graphs.run
    def timed_run(key, fn, run=run, tag=tag):
        t0 = time.perf_counter()
        r = run(key, fn)
        T[f"{tag}.run {key[0]}"] = T.get(f"{tag}.run {key[0]}", 0.0) + time.perf_counter() - t0
        return r
    m._graphs.run = timed_run
lib = _lib.lib()
orig_sync = lib.dcvc_stream_sync
pkts = []
for i, x in enumerate(frames):
    if i == 4:
        T.clear(); t_enc = t_dec = 0.0; n = 0
    t0 = time.perf_counter(); p = enc.encode(x); torch.cuda.synchronize(); t1 = time.perf_counter()
    xh = dec.decode(p); torch.cuda.synchronize(); t2 = time.perf_counter()
    if i >= 4:
        t_enc += t1 - t0; t_dec += t2 - t1; n += 1
print(f"enc {1e3*t_enc/n:.2f} ms/frame   dec {1e3*t_dec/n:.2f} ms/frame")
for k in sorted(T):
    print(f"  {k:34s} {1e3*T[k]/n:7.3f} ms/frame")
