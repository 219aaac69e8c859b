"""Generates the golden fixtures in this directory by RUNNING THE REFERENCE in the build container
(torch CPU fallback ops + the reference's own rANS coder compiled into oracle/_ref).

    python tests/golden/make_golden.py [--skip-1080p]

The reference has no tests, golden vectors or checkpoints of its own, so
these fixtures are what pins parity: weights come from opendcvc_amd.weights (numpy PCG64, seed in
each fixture) loaded into the reference modules with load_state_dict; inputs from
opendcvc_amd.weights.synthetic_frame_yuv444.  Only data (inputs / expected outputs / hashes) is
written - no reference source.  The GPU box never runs this script.
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)

import ref_harness  # noqa: E402
from opendcvc_amd import weights  # noqa: E402

SEED = 1234
THRES = 0.12
INDEX_MAP = [0, 1, 0, 2, 0, 2, 0, 2]


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def t2n(t):
    return t.detach().cpu().numpy()


def load_models(DMC, DMCI):
    sdi = weights.make_state_dict("dmci", SEED)
    sdp = weights.make_state_dict("dmc", SEED)
    i_net = DMCI().eval()
    i_net.load_state_dict({k: torch.from_numpy(v) for k, v in sdi.items()})
    i_net.update(THRES)
    p_net = DMC().eval()
    p_net.load_state_dict({k: torch.from_numpy(v) for k, v in sdp.items()})
    p_net.update(THRES)
    return i_net, p_net


def gen_rans(R):
    rng = np.random.default_rng(7)
    ncdf, maxlen = 128, 19
    cdf = np.zeros((ncdf, maxlen + 2), np.int32)
    sizes = np.zeros(ncdf, np.int32)
    offs = np.zeros(ncdf, np.int32)
    pmf_in, pmf_out = [], []
    for i in range(ncdf):
        n = int(rng.integers(3, maxlen + 1))
        p = rng.random(n).astype(np.float32) ** 3
        p /= p.sum()
        c = np.array(R.pmf_to_quantized_cdf(p.tolist(), 16), np.int64)
        if i < 8:
            pmf_in.append(np.pad(p, (0, maxlen - n)))
            pmf_out.append(np.pad(c, (0, maxlen + 1 - c.size)))
        cdf[i, :n + 1] = c
        sizes[i] = n + 1
        offs[i] = -(n // 2)
    out = dict(cdf=cdf, sizes=sizes, offsets=offs, pmf_in=np.array(pmf_in, np.float32),
               pmf_out=np.array(pmf_out, np.int64), pmf_len=np.array([int(s) - 1 for s in sizes[:8]]))
    n = 4001
    idx = rng.integers(0, ncdf, n).astype(np.uint8)
    sym = np.clip(np.round(rng.standard_normal(n) * 3), -128, 127).astype(np.int16)
    sym[::97] = rng.integers(-128, 128, sym[::97].size)          # far escapes
    packed = ((sym.astype(np.int32) << 8) + idx).astype(np.int16)
    z = np.clip(np.round(rng.standard_normal(128 * 6) * 4), -128, 127).astype(np.int8)
    out.update(idx=idx, sym=sym, z=z)
    for two in (0, 1):
        enc = R.RansEncoder()
        enc.add_cdf(cdf, sizes, offs)
        enc.set_use_two_encoders(bool(two))
        enc.reset()
        enc.encode_z(z, 0, 0, 6)
        enc.encode_y(packed, 0)
        enc.encode_y(packed[:777], 0)
        enc.encode_y(packed[:0], 0)                               # zero-length task
        enc.flush()
        out[f"stream_two{two}"] = np.array(enc.get_encoded_stream(), np.uint8)
        # short y-only inputs (the reference sizes its scratch buffer at one byte per symbol,
        # rans.cpp:221, so very short or escape-heavy inputs overflow it there; stay clear of that)
        small = ((np.clip(sym, -2, 2).astype(np.int32) << 8) + idx).astype(np.int16)
        for m in (64, 201):
            enc.reset()
            enc.encode_y(small[:m], 0)
            enc.flush()
            out[f"stream_two{two}_y{m}"] = np.array(enc.get_encoded_stream(), np.uint8)
    np.savez_compressed(os.path.join(HERE, "rans_kat.npz"), **out)


def gen_tables(i_net, p_net):
    g = i_net.gaussian_encoder.get_cdf_info()
    np.savez_compressed(os.path.join(HERE, "gauss_cdf.npz"), cdf=g[0], length=g[1], offset=g[2])
    meta = {}
    for name, net in (("dmci", i_net), ("dmc", p_net)):
        c, l, o = net.bit_estimator_z.get_cdf_info()
        c = np.ascontiguousarray(c, np.int32)
        meta[name] = dict(shape=list(c.shape), cdf_sha256=sha(c.tobytes()),
                          length_sha256=sha(np.ascontiguousarray(l, np.int32).tobytes()),
                          offset_sha256=sha(np.ascontiguousarray(o, np.int32).tobytes()),
                          first_rows=c[:4].tolist(), seed=SEED)
    json.dump(meta, open(os.path.join(HERE, "ztables.json"), "w"), indent=1)


def gen_ops(L, O):
    """Per-operator vectors from the reference layer classes / fallback ops at tiny shapes."""
    rng = np.random.default_rng(11)
    out = {}

    def rnd(*s):
        return rng.standard_normal(s).astype(np.float32)

    def init(mod):
        for _, p in mod.named_parameters():
            p.data = torch.from_numpy(rnd(*p.shape) * np.float32(0.5 / np.sqrt(max(1, np.prod(p.shape[1:])))))
        return mod

    def dump_mod(prefix, mod):
        for k, v in mod.state_dict().items():
            out[f"{prefix}.w.{k}"] = t2n(v)

    H, W = 6, 10
    cases = [("dcb_plain", 16, 16, False, False, False), ("dcb_adapt", 24, 16, False, False, False),
             ("dcb_short", 16, 16, True, False, False), ("dcb_quant", 16, 16, False, True, False),
             ("dcb_force", 16, 16, False, False, True), ("dcb_adapt_short_q", 40, 32, True, True, False)]
    for name, cin, c, short, quant, force in cases:
        m = init(L.DepthConvBlock(cin, c, shortcut=short, force_adaptor=force)).eval()
        x = torch.from_numpy(rnd(1, cin, H, W))
        q = torch.from_numpy(rng.uniform(0.5, 1.5, (1, c, 1, 1)).astype(np.float32)) if quant else None
        y = m.forward_torch(x, quant_step=q)
        dump_mod(name, m)
        out[name + ".x"] = t2n(x)
        out[name + ".y"] = t2n(y)
        if q is not None:
            out[name + ".q"] = t2n(q)
    for name, k in (("subpel1", 1), ("subpel3", 3)):
        m = init(L.SubpelConv2x(16, 8, k, padding=k // 2)).eval()
        x = torch.from_numpy(rnd(1, 16, H, W))
        dump_mod(name, m)
        out[name + ".x"] = t2n(x)
        out[name + ".y"] = t2n(m.forward_torch(x))
    m = init(L.ResidualBlockWithStride2(16, 24)).eval()
    x = torch.from_numpy(rnd(1, 16, H, W))
    dump_mod("resdown", m)
    out["resdown.x"] = t2n(x)
    out["resdown.y"] = t2n(m(x))
    m = init(L.ResidualBlockUpsample(16, 8)).eval()
    dump_mod("resup", m)
    out["resup.x"] = t2n(x)
    out["resup.y"] = t2n(m(x))
    m = init(torch.nn.Conv2d(16, 12, 3, stride=2, padding=1)).eval()
    dump_mod("conv3s2", m)
    out["conv3s2.x"] = t2n(x)
    out["conv3s2.y"] = t2n(m(x))

    # elementwise entropy glue (cuda_inference.py fallbacks)
    C = 8
    y = torch.from_numpy(rnd(1, C, H, W) * 3)
    sc = torch.from_numpy(np.abs(rnd(1, C, H, W)) * 0.6)
    mu = torch.from_numpy(rnd(1, C, H, W))
    mask = torch.from_numpy((rng.random((1, C, H, W)) > 0.5).astype(np.float32))
    r = O.process_with_mask(y, sc, mu, mask, 0.12)
    out.update({"pwm.y": t2n(y), "pwm.scales": t2n(sc), "pwm.means": t2n(mu), "pwm.mask": t2n(mask)})
    for n_, v in zip(("y_res", "y_q", "y_hat", "s_hat"), r):
        out["pwm." + n_] = t2n(v)
    smin, smax = 0.11, 16.0
    lmin = float(np.log(smin))
    lrec = 1.0 / ((np.log(smax) - np.log(smin)) / 127)
    scales = torch.from_numpy(np.exp(rng.uniform(np.log(0.05), np.log(20.0), 4096)).astype(np.float32))
    syms = torch.from_numpy(np.round(rng.standard_normal(4096) * 3).astype(np.float32))
    out["idx.scales"] = t2n(scales)
    out["idx.symbols"] = t2n(syms)
    idx, cond = O.build_index_dec(scales.clone(), smin, smax, lmin, lrec, 0.12)
    out["idx.dec_idx"] = t2n(idx)
    out["idx.dec_cond"] = t2n(cond)
    out["idx.enc_packed"] = t2n(O.build_index_enc(syms, scales.clone(), smin, smax, lmin, lrec, 0.12))
    z = torch.from_numpy(rnd(1, 4, 3, 5) * 60)
    zh, z8 = O.round_and_to_int8(z)
    out.update({"z.in": t2n(z), "z.hat": t2n(zh), "z.int8": t2n(z8)})
    qd = torch.from_numpy(rnd(1, C, H, W) + 1)
    qc, yq = O.clamp_reciprocal_with_quant(qd, y, 0.5)
    out.update({"crq.q": t2n(qd), "crq.q_out": t2n(qc), "crq.y_out": t2n(yq)})
    x3 = torch.from_numpy(rnd(1, 3, 5, 7))
    out.update({"pad.x": t2n(x3), "pad.y": t2n(O.replicate_pad(x3, 3, 9))})
    xs = torch.from_numpy(rnd(1, 192, 2, 3))
    bs = torch.from_numpy(rnd(192))
    out.update({"ps8.x": t2n(xs), "ps8.b": t2n(bs), "ps8.y": t2n(O.bias_pixel_shuffle_8(xs, bs))})
    np.savez_compressed(os.path.join(HERE, "ops_small.npz"), **out)


def gen_container(S):
    """Container KATs from the reference's stream_helper (stream_helper.py:68-217)."""
    import io
    out = {"varint": [], "sps": [], "ip": []}
    for v in (0, 1, 127, 128, 129, 255, 256, 16383, 16384, 65535, 70000, (1 << 24) + 5, (1 << 30) - 1):
        f = io.BytesIO()
        n = S.write_uint_adaptive(f, v)
        out["varint"].append({"value": v, "hex": f.getvalue().hex(), "n": n})
    for sps in ({"sps_id": 0, "height": 1080, "width": 1920, "ec_part": 1, "use_ada_i": 0},
                {"sps_id": 3, "height": 64, "width": 64, "ec_part": 0, "use_ada_i": 1},
                {"sps_id": 15, "height": 2160, "width": 3840, "ec_part": 1, "use_ada_i": 1},
                {"sps_id": 1, "height": 20000, "width": 100, "ec_part": 0, "use_ada_i": 0}):
        f = io.BytesIO()
        n = S.write_sps(f, sps)
        out["sps"].append({"sps": sps, "hex": f.getvalue().hex(), "n": n})
    rng = np.random.default_rng(5)
    for is_i, sps_id, qp, ln in ((True, 0, 0, 0), (False, 2, 63, 5), (False, 15, 255, 130), (True, 1, 32, 20000)):
        payload = rng.integers(0, 256, ln, dtype=np.uint8).tobytes()
        f = io.BytesIO()
        n = S.write_ip(f, is_i, sps_id, qp, payload)
        out["ip"].append({"is_i": is_i, "sps_id": sps_id, "qp": qp, "payload_sha256": sha(payload), "payload_len": ln,
                          "hex_head": f.getvalue()[:8].hex(), "sha256": sha(f.getvalue()), "n": n})
    # a whole multi-frame stream with SPS dedup, as test_video.py:216-224 writes it
    f = io.BytesIO()
    helper = S.SPSHelper()
    frames = [(True, 32, 0, 300), (False, 40, 1, 90), (False, 32, 0, 70), (False, 36, 0, 16500), (False, 32, 1, 10)]
    spec = []
    for is_i, qp, ada, ln in frames:
        payload = rng.integers(0, 256, ln, dtype=np.uint8).tobytes()
        sps = {"sps_id": -1, "height": 1080, "width": 1920, "ec_part": 1, "use_ada_i": ada}
        sps_id, new = helper.get_sps_id(sps)
        sps["sps_id"] = sps_id
        if new:
            S.write_sps(f, sps)
        S.write_ip(f, is_i, sps_id, qp, payload)
        spec.append({"is_i": is_i, "qp": qp, "use_ada_i": ada, "payload_hex": payload.hex() if ln < 400 else None,
                     "payload_sha256": sha(payload), "payload_len": ln})
    out["stream"] = {"frames": spec, "sha256": sha(f.getvalue()), "n": len(f.getvalue()), "seed": 5}
    json.dump(out, open(os.path.join(HERE, "container_kat.json"), "w"), indent=1)


def run_sequence(i_net, p_net, h, w, n_frames, qp, two, reset_interval, keep_tensors):
    """Encode then decode a short sequence the way test_video.py:164-214,258-285 drives the models."""
    rec = dict(h=h, w=w, qp=qp, two=int(two), reset_interval=reset_interval, seed=SEED, thres=THRES,
               frames=[])
    tensors = {}
    i_net.set_use_two_entropy_coders(two)
    p_net.set_use_two_entropy_coders(two)
    p_net.set_curr_poc(0)
    last_qp = 0
    streams = []
    for fi in range(n_frames):
        x = torch.from_numpy(weights.synthetic_frame_yuv444(h, w, fi, 0))
        use_ada_i = 0
        if fi == 0:
            cur_qp = qp
            enc = i_net.compress(x, qp)
            p_net.clear_dpb()
            p_net.add_ref_frame(None, enc["x_hat"])
        else:
            fa = INDEX_MAP[fi % 8]
            if reset_interval > 0 and fi % reset_interval == 1:
                use_ada_i = 1
                p_net.prepare_feature_adaptor_i(last_qp)
            cur_qp = p_net.shift_qp(qp, fa)
            enc = p_net.compress(x, cur_qp)
            last_qp = cur_qp
            if keep_tensors:
                tensors[f"enc_feature_{fi}"] = t2n(p_net.dpb[0].feature)
        streams.append((fi == 0, cur_qp, use_ada_i, enc["bit_stream"]))
        rec["frames"].append(dict(type="I" if fi == 0 else "P", qp=cur_qp, use_ada_i=use_ada_i,
                                  bytes=len(enc["bit_stream"]), sha256=sha(enc["bit_stream"])))
        if keep_tensors:
            tensors[f"stream_{fi}"] = np.frombuffer(enc["bit_stream"], np.uint8)
    # decode
    p_net.set_curr_poc(0)
    for fi, (is_i, cur_qp, use_ada_i, bits) in enumerate(streams):
        sps = dict(height=h, width=w, ec_part=1 if two else 0, use_ada_i=use_ada_i)
        if is_i:
            dec = i_net.decompress(bits, sps, cur_qp)
            p_net.clear_dpb()
            p_net.add_ref_frame(None, dec["x_hat"])
        else:
            if use_ada_i:
                p_net.reset_ref_feature()
            dec = p_net.decompress(bits, sps, cur_qp)
        xh = t2n(dec["x_hat"])
        x = weights.synthetic_frame_yuv444(h, w, fi, 0)
        f = rec["frames"][fi]
        f["psnr"] = float(-10 * np.log10(np.mean((xh - x) ** 2)))
        f["x_hat_mean"] = float(xh.mean())
        f["x_hat_u8_sha256"] = sha(np.clip(np.round(xh * 255), 0, 255).astype(np.uint8).tobytes())
        if keep_tensors:
            tensors[f"x_hat_{fi}"] = xh.astype(np.float32)
    return rec, tensors


def main():
    DMC, DMCI, L, O, R, S = ref_harness.load()
    torch.set_grad_enabled(False)
    torch.manual_seed(0)
    gen_rans(R)
    gen_container(S)
    gen_ops(L, O)
    i_net, p_net = load_models(DMC, DMCI)
    gen_tables(i_net, p_net)
    rec, tens = run_sequence(i_net, p_net, 64, 64, 7, 32, False, 4, True)
    np.savez_compressed(os.path.join(HERE, "seq_64.npz"), meta=json.dumps(rec), **tens)
    out = {"seq_64": rec}
    out["seq_64_two"] = run_sequence(i_net, p_net, 64, 64, 3, 21, True, 0, False)[0]
    out["seq_80x48"] = run_sequence(i_net, p_net, 48, 80, 3, 40, False, 0, False)[0]   # ragged: y 3x5, z 1x2
    out["seq_256"] = run_sequence(i_net, p_net, 256, 256, 4, 32, False, 0, False)[0]  # config 0 + P frames
    if "--skip-1080p" not in sys.argv:
        # padded 1088x1920 like test_video.py:150,179 would feed the models
        out["seq_1088x1920"] = run_sequence(i_net, p_net, 1088, 1920, 3, 32, True, 0, False)[0]
    json.dump(out, open(os.path.join(HERE, "sequences.json"), "w"), indent=1)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
