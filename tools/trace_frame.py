"""Developer tool: sequential host trace of one 1080p P-frame encode and decode.  Every wrapped call
prints when the host entered / left it (ms since the frame started); a CUDA event recorded at each
exit gives the time the GPU reached that point, so GPU-idle and host-blocked intervals are visible."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from opendcvc_amd.pipeline import SequenceDecoder, SequenceEncoder

torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
(ie, pe), (idec, pdec) = bench.load_models(torch.float16, dev, 1, 0)
for m in (ie, pe, idec, pdec):
    m.set_use_two_entropy_coders(True)
frames = bench.make_frames(0, torch.float16, dev)[:10]
enc = SequenceEncoder(ie, pe, 32, intra_period=32, reset_interval=32)
dec = SequenceDecoder(idec, pdec, 1080, 1920, True)

LOG = []
ON = [False]
T0 = [0.0]
E0 = [None]


def wrap(obj, name, label):
    f = getattr(obj, name)

    def w(*a, **k):
        if not ON[0]:
            return f(*a, **k)
        t0 = time.perf_counter()
        r = f(*a, **k)
        t1 = time.perf_counter()
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        LOG.append((label, t0 - T0[0], t1 - T0[0], ev))
        return r
    setattr(obj, name, w)


def wrap_runs(model, tag):
    """label every captured run (GraphCache.run) by its name"""
    f = model._graphs.run

    def w(key, fn):
        if not ON[0]:
            return f(key, fn)
        t0 = time.perf_counter()
        r = f(key, fn)
        t1 = time.perf_counter()
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        LOG.append((f"{tag}.run {key[0]}", t0 - T0[0], t1 - T0[0], ev))
        return r
    model._graphs.run = w


for m, tag in ((pe, "enc"), (pdec, "dec")):
    ec = m.entropy_coder
    for n in ("encode_y", "encode_z", "flush", "get_encoded_stream", "set_stream", "decode_z", "decode_and_get_y",
              "get_decoded", "reset"):
        wrap(ec, n, f"{tag}.ec.{n}")
    for n in ("_stage_q", "_stage_reference", "_unshuffle8"):
        wrap(m, n, f"{tag}.{n}")
    wrap_runs(m, tag)
_es = torch.cuda.Event.synchronize


def ev_sync(self):
    if not ON[0]:
        return _es(self)
    t0 = time.perf_counter()
    _es(self)
    LOG.append(("event.synchronize", t0 - T0[0], time.perf_counter() - T0[0], None))


torch.cuda.Event.synchronize = ev_sync


def dump(title, t_end):
    print(f"--- {title}: {1e3 * t_end:.3f} ms")
    for label, a, b, ev in LOG:
        g = f"gpu@{E0[0].elapsed_time(ev):7.3f}" if ev is not None else ""
        print(f"  {1e3 * a:7.3f} -> {1e3 * b:7.3f}  ({1e3 * (b - a):6.3f})  {label:30s} {g}")
    LOG.clear()


for i, x in enumerate(frames):
    trace = i == 6
    torch.cuda.synchronize()
    ON[0] = trace
    E0[0] = torch.cuda.Event(enable_timing=True)
    E0[0].record()
    T0[0] = time.perf_counter()
    p = enc.encode(x)
    torch.cuda.synchronize()
    t1 = time.perf_counter() - T0[0]
    if trace:
        dump("encode", t1)
    E0[0] = torch.cuda.Event(enable_timing=True)
    E0[0].record()
    T0[0] = time.perf_counter()
    xh = dec.decode(p)
    torch.cuda.synchronize()
    t2 = time.perf_counter() - T0[0]
    if trace:
        dump("decode", t2)
