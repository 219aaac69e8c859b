"""Drives a pair of (intra, inter) codecs through a short sequence exactly like the reference
harness does (test_video.py:164-214 encode loop, :258-285 decode loop).  Works for the oracle
codecs and for the HIP codecs (same method names)."""
import hashlib

import numpy as np

from opendcvc_amd import weights

INDEX_MAP = [0, 1, 0, 2, 0, 2, 0, 2]


def run_sequence(i_net, p_net, rec, n_frames=None, to_x=lambda a: a, to_np=lambda a: a, feature_of=None):
    two = bool(rec["two"])
    i_net.set_use_two_entropy_coders(two)
    p_net.set_use_two_entropy_coders(two)
    h, w, qp = rec["h"], rec["w"], rec["qp"]
    frames = rec["frames"] if n_frames is None else rec["frames"][:n_frames]
    out = []
    last_qp = 0
    p_net.set_curr_poc(0) if hasattr(p_net, "set_curr_poc") else None
    for fi, f in enumerate(frames):
        x = to_x(weights.synthetic_frame_yuv444(h, w, fi, 0))
        if fi == 0:
            enc = i_net.compress(x, qp)
            p_net.clear_dpb()
            p_net.add_ref_frame(None, enc["x_hat"])
        else:
            if rec["reset_interval"] > 0 and fi % rec["reset_interval"] == 1:
                p_net.prepare_feature_adaptor_i(last_qp)
            cur = p_net.shift_qp(qp, INDEX_MAP[fi % 8])
            assert cur == f["qp"]
            enc = p_net.compress(x, cur)
            last_qp = cur
        out.append(dict(bits=enc["bit_stream"],
                        feature=None if (fi == 0 or feature_of is None) else feature_of(p_net)))
    p_net.clear_dpb()
    p_net.set_curr_poc(0) if hasattr(p_net, "set_curr_poc") else None
    for fi, f in enumerate(frames):
        sps = dict(height=h, width=w, ec_part=rec["two"], use_ada_i=f["use_ada_i"])
        if fi == 0:
            dec = i_net.decompress(out[fi]["bits"], sps, f["qp"])
            p_net.clear_dpb()
            p_net.add_ref_frame(None, dec["x_hat"])
        else:
            if f["use_ada_i"]:
                p_net.reset_ref_feature()
            dec = p_net.decompress(out[fi]["bits"], sps, f["qp"])
        out[fi]["x_hat"] = to_np(dec["x_hat"])
        out[fi]["dec_feature"] = None if (fi == 0 or feature_of is None) else feature_of(p_net)
    return out


def psnr_of(rec, fi, x_hat):
    x = weights.synthetic_frame_yuv444(rec["h"], rec["w"], fi, 0)
    return float(-10 * np.log10(np.mean((np.asarray(x_hat, np.float64) - x) ** 2)))


def check_against_record(rec, got, min_exact=1.0, tol=1e-4):
    """fp32 parity bar against the reference (BASELINE.json: PSNR / bpp within 1e-4): every frame's
    stream length within `tol` relative (= bpp within tol relative) and PSNR within `tol` dB; at
    least `min_exact` of the frames byte-identical (a different fp32 summation order can move a
    value across a rounding boundary and flip one symbol in a long stream)."""
    exact = 0
    for fi, (f, g) in enumerate(zip(rec["frames"], got)):
        assert abs(len(g["bits"]) - f["bytes"]) <= max(1, tol * f["bytes"]), f"frame {fi}: stream length"
        exact += hashlib.sha256(g["bits"]).hexdigest() == f["sha256"]
        psnr = psnr_of(rec, fi, g["x_hat"])
        assert abs(psnr - f["psnr"]) < tol, f"frame {fi}: psnr {psnr} vs {f['psnr']}"
    assert exact >= min_exact * len(got), f"only {exact}/{len(got)} streams byte-identical"
