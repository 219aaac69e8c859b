"""Developer micro-benchmark: the host entropy coder alone on the real inputs of one 1080p P frame
(tests/golden/coder_inputs_1080p.npz, dumped on the GPU box by tools/dump_coder_inputs.py).
Runs the exact call sequence DMC.compress / DMC.decompress make, checks the round trip and that the
stream equals the recorded one, and prints per-call wall times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from opendcvc_amd.models import DMC
from opendcvc_amd import weights

d = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "coder_inputs_1080p.npz"))
m = DMC()
m.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in weights.make_state_dict("dmc", 1234).items()})
m.update(0.12)
m.set_use_two_entropy_coders(True)
ec = m.entropy_coder
z, (zg, zoff, zper) = d["z"], d["z_args"]
p0, p1, i0, i1 = d["packed0"], d["packed1"], d["index0"], d["index1"]
yg = int(d["y_group"])
n = p0.size
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
T = {}


def tm(label, f, *a):
    t0 = time.perf_counter()
    r = f(*a)
    T.setdefault(label, []).append(time.perf_counter() - t0)
    return r


out0 = np.empty(n, np.int8); out1 = np.empty(n, np.int8); zo = np.empty(z.size, np.int8)
c0, c1 = np.ascontiguousarray(i0[i0 != 0xff]), np.ascontiguousarray(i1[i1 != 0xff])
co0, co1 = np.empty(c0.size, np.int8), np.empty(c1.size, np.int8)
for r in range(reps):
    tm("enc.reset", ec.reset)
    tm("enc.encode_z", ec.encode_z, z, int(zg), int(zoff), int(zper))
    tm("enc.encode_y0", ec.encode_y, p0, yg, True)
    tm("enc.encode_y1", ec.encode_y, p1, yg, True)
    tm("enc.flush", ec.flush)
    s = tm("enc.get_stream", ec.get_encoded_stream)
    tm("dec.set_stream", ec.set_stream, s)
    tm("dec.decode_z", ec.decode_z, z.size, int(zg), int(zoff), int(zper))
    tm("dec.get_z", ec.get_decoded, zo)
    tm("dec.step0", ec.decode_and_get_y, i0, yg, out0)
    tm("dec.step1", ec.decode_and_get_y, i1, yg, out1)
    # the same two steps in the compacted form (what DMC.decompress uses: the kept indexes only)
    tm("dec.set_stream", ec.set_stream, s)
    ec.decode_z(z.size, int(zg), int(zoff), int(zper)); ec.get_decoded(zo)
    tm("dec.compact0", ec.decode_compact, c0, c0.size, yg, co0)
    tm("dec.compact1", ec.decode_compact, c1, c1.size, yg, co1)
assert np.array_equal(np.frombuffer(s, np.uint8), d["stream"]), "stream differs from the recorded one"
assert np.array_equal(zo, z)
for p, i, o in ((p0, i0, out0), (p1, i1, out1)):
    kept = (p & 0xff) != 0xff
    assert np.array_equal(kept, i != 0xff) and np.array_equal((p >> 8)[kept].astype(np.int8), o[kept]) and not o[~kept].any()
assert np.array_equal(co0, out0[i0 != 0xff]) and np.array_equal(co1, out1[i1 != 0xff])
kept0, kept1 = int(((p0 & 0xff) != 0xff).sum()), int(((p1 & 0xff) != 0xff).sum())
print(f"kept symbols: step0 {kept0}  step1 {kept1}  of {n} each; z {z.size}; stream {len(s)} bytes")
enc = dec = 0.0
for k, v in T.items():
    med = 1e3 * float(np.median(v))
    print(f"  {k:16s} {med:7.3f} ms")
    if k.startswith("enc"):
        enc += med
    else:
        dec += med
print(f"  encode total {enc:.3f} ms   decode total {dec:.3f} ms")
