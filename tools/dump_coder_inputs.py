"""Developer tool: save the host entropy coder's real inputs for one 1080p P frame (z symbols, packed y
symbols of both checkerboard steps, the decoder's cdf-index arrays, the stream) so tools/rans_bench.py
can time the coder alone on any machine."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from opendcvc_amd.pipeline import SequenceDecoder, SequenceEncoder

torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
(ie, pe), (idec, pdec) = bench.load_models(torch.float16, dev, 1, 0)
for m in (ie, pe, idec, pdec):
    m.set_use_two_entropy_coders(True)
frames = bench.make_frames(0, torch.float16, dev)[:6]
enc = SequenceEncoder(ie, pe, 32, intra_period=32, reset_interval=32)
dec = SequenceDecoder(idec, pdec, 1080, 1920, True)
REC = {}
ON = [False]


def tap(ec, name, fn):
    f = getattr(ec, name)

    def w(*a, **k):
        if ON[0]:
            fn(*a, **k)
        return f(*a, **k)
    setattr(ec, name, w)


def rec_list(key):
    def fn(*a, **k):
        REC.setdefault(key, []).append([np.array(v).copy() if isinstance(v, np.ndarray) else v for v in a])
    return fn


tap(pe.entropy_coder, "encode_z", rec_list("encode_z"))
tap(pe.entropy_coder, "encode_y", rec_list("encode_y"))
tap(pdec.entropy_coder, "decode_z", rec_list("decode_z"))
tap(pdec.entropy_coder, "decode_and_get_y", rec_list("decode_y"))
for i, x in enumerate(frames):
    ON[0] = i == 4
    p = enc.encode(x)
    if ON[0]:
        REC["stream"] = np.frombuffer(p.bit_stream, np.uint8).copy()
    dec.decode(p)
out = {"stream": REC["stream"]}
z, zg, zoff, zper = REC["encode_z"][0]
out.update(z=z, z_args=np.array([zg, zoff, zper]))
for k, (sym, g) in enumerate(REC["encode_y"]):
    out[f"packed{k}"] = sym
    out["y_group"] = np.array(g)
out["decode_z_args"] = np.array(REC["decode_z"][0])
for k, (idx, g, o) in enumerate(REC["decode_y"]):
    out[f"index{k}"] = idx
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed("gpurun_out/coder_inputs.npz", **out)
print({k: (v.shape, v.dtype) for k, v in out.items()})
