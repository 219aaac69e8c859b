"""The product's host rANS coder (C ABI of libdcvc_amd.so; pure host code, runs without a GPU) against
the reference's golden streams (bit-exact) and against the oracle on random inputs, plus the sentinel
convention that replaces the reference's boolean-mask compaction."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

import dcvc_oracle as O
from opendcvc_amd import _lib

pytestmark = pytest.mark.skipif(not os.path.exists(_lib.LIB_PATH), reason="libdcvc_amd.so not built")


@pytest.fixture(scope="module")
def kat(golden_dir):
    return np.load(os.path.join(golden_dir, "rans_kat.npz"))


def coder(k, two):
    from opendcvc_amd.entropy import EntropyCoder
    c = EntropyCoder()
    assert c.add_cdf(k["cdf"], k["sizes"], k["offsets"]) == 0
    c.set_use_two_entropy_coders(two)
    return c


def packed_of(k):
    return ((k["sym"].astype(np.int32) << 8) + k["idx"]).astype(np.int16)


@pytest.mark.parametrize("two", [0, 1])
def test_encode_matches_reference_stream(kat, two):
    c = coder(kat, two)
    p = packed_of(kat)
    c.reset()
    c.encode_z(kat["z"], 0, 0, 6)
    c.encode_y(p, 0)
    c.encode_y(p[:777], 0)
    c.encode_y(p[:0], 0)
    c.flush()
    assert c.get_encoded_stream() == kat[f"stream_two{two}"].tobytes()
    small = ((np.clip(kat["sym"], -2, 2).astype(np.int32) << 8) + kat["idx"]).astype(np.int16)
    for m in (64, 201):
        c.reset()
        c.encode_y(small[:m], 0)
        c.flush()
        assert c.get_encoded_stream() == kat[f"stream_two{two}_y{m}"].tobytes()


@pytest.mark.parametrize("two", [0, 1])
def test_decode_reference_stream(kat, two):
    c = coder(kat, two)
    c.set_stream(kat[f"stream_two{two}"].tobytes())
    z = np.empty(kat["z"].size, np.int8)
    c.decode_z(z.size, 0, 0, 6)
    c.get_decoded(z)
    assert np.array_equal(z, kat["z"])
    y = np.empty(kat["idx"].size, np.int8)
    c.decode_and_get_y(kat["idx"], 0, y)
    assert np.array_equal(y, kat["sym"].astype(np.int8))
    y2 = np.empty(777, np.int8)
    c.decode_and_get_y(kat["idx"][:777], 0, y2)
    assert np.array_equal(y2, kat["sym"][:777].astype(np.int8))


@pytest.mark.parametrize("two", [0, 1])
def test_sentinel_equals_compaction_and_roundtrips(two):
    """Fixed-size arrays with 0xFF sentinels == the reference's compacted arrays, vs the oracle."""
    g = O.gaussian_tables()
    rng = np.random.default_rng(5)
    n = 30011
    idx = rng.integers(0, 128, n).astype(np.uint8)
    sigma = 0.11 * (16 / 0.11) ** (idx / 127.0)
    sym = np.clip(np.round(rng.standard_normal(n) * sigma), -128, 127).astype(np.int16)
    keep = rng.random(n) < 0.3
    full = ((sym.astype(np.int32) << 8) + np.where(keep, idx, 0xFF)).astype(np.int16)
    compact = ((sym.astype(np.int32) << 8) + idx).astype(np.int16)[keep]
    from opendcvc_amd.entropy import EntropyCoder
    c = EntropyCoder()
    c.add_cdf(*g)
    c.set_use_two_entropy_coders(two)
    c.reset()
    c.encode_y(full, 0)
    c.flush()
    mine = c.get_encoded_stream()
    o = O.Coder()
    o.add_cdf(*g)
    o.set_use_two(two)
    o.reset()
    o.encode_y(compact, 0)
    assert mine == o.flush()
    c.set_stream(mine)
    out = np.empty(n, np.int8)
    c.decode_and_get_y(np.where(keep, idx, 0xFF).astype(np.uint8), 0, out)
    assert np.array_equal(out[keep], sym[keep].astype(np.int8)) and not np.any(out[~keep])


@pytest.mark.parametrize("two", [0, 1])
@pytest.mark.parametrize("kept_frac", [0.0, 0.04, 0.5, 1.0])
def test_compact_decode_equals_sentinel_decode(two, kept_frac):
    """dcvc_rans_dec_decode_compact (kept indexes only, as the device compaction hands them over) consumes a stream exactly
    like decode_and_get_y on the sentinel array: same symbols, same coder split, same end state - over two consecutive
    calls on one stream (the two checkerboard steps of a frame) and for 0 / 1 / odd counts."""
    from opendcvc_amd.entropy import EntropyCoder
    g = O.gaussian_tables()
    rng = np.random.default_rng(17 + two)
    steps = []
    for n in (3, 30011, 1, 8193):
        idx = rng.integers(0, 128, n).astype(np.uint8)
        sigma = 0.11 * (16 / 0.11) ** (idx / 127.0)
        sym = np.clip(np.round(rng.standard_normal(n) * sigma), -128, 127).astype(np.int16)
        keep = rng.random(n) < (1.0 if n == 3 else kept_frac)        # (the first step keeps all: the stream is never empty)
        steps.append((np.where(keep, idx, 0xFF).astype(np.uint8), keep, sym))
    c = EntropyCoder()
    c.add_cdf(*g)
    c.set_use_two_entropy_coders(bool(two))
    c.reset()
    for full, keep, sym in steps:
        c.encode_y(((sym.astype(np.int32) * 256 + full) & 0xFFFF).astype(np.uint16).view(np.int16), 0)
    c.flush()
    stream = c.get_encoded_stream()
    c.set_stream(stream)
    want = []
    for full, keep, sym in steps:
        out = np.empty(full.size, np.int8)
        c.decode_and_get_y(full, 0, out)
        want.append(out)
    c.check_end()
    c.set_stream(stream)
    for (full, keep, sym), w in zip(steps, want):
        cidx = np.concatenate([full[keep], np.full(5, 0xAB, np.uint8)])      # (entries behind `count` are never looked at)
        out = np.full(cidx.size, 77, np.int8)
        assert c.decode_compact(cidx, int(keep.sum()), 0, out) == int(keep.sum())
        assert np.array_equal(out[:int(keep.sum())], w[keep]) and np.all(out[int(keep.sum()):] == 77)
        assert np.array_equal(w[keep], sym[keep].astype(np.int8))
    c.check_end()
    with pytest.raises(_lib.DcvcError):
        c.decode_compact(np.zeros(3, np.uint8), 4, 0, np.zeros(8, np.int8))          # count beyond the array
    with pytest.raises(_lib.DcvcError):
        c.decode_compact(np.full(3, 200, np.uint8), 3, 0, np.zeros(8, np.int8))      # table index out of range


def test_pmf_to_quantized_cdf(kat):
    from opendcvc_amd.entropy import pmf_to_quantized_cdf
    for p, c, n in zip(kat["pmf_in"], kat["pmf_out"], kat["pmf_len"]):
        assert np.array_equal(pmf_to_quantized_cdf(p[:n]).astype(np.int64), c[:n + 1])


def test_tables_match_reference(golden_dir):
    from opendcvc_amd import entropy, weights
    g = np.load(os.path.join(golden_dir, "gauss_cdf.npz"))
    cdf, length, offset = entropy.gaussian_cdf_tables()
    assert np.array_equal(cdf, g["cdf"]) and np.array_equal(length, g["length"]) and np.array_equal(offset, g["offset"])
    meta = json.load(open(os.path.join(golden_dir, "ztables.json")))
    for model, qp_num in (("dmci", 64), ("dmc", 72)):
        sd = weights.make_state_dict(model, meta[model]["seed"])
        pre = "bit_estimator_z."
        params = {k[len(pre):]: torch.from_numpy(v) for k, v in sd.items() if k.startswith(pre)}
        cdf, length, offset = entropy.factorized_cdf_tables(params, qp_num, 128)
        assert hashlib.sha256(np.ascontiguousarray(cdf, np.int32).tobytes()).hexdigest() == meta[model]["cdf_sha256"]
        assert hashlib.sha256(length.tobytes()).hexdigest() == meta[model]["length_sha256"]
        assert hashlib.sha256(offset.tobytes()).hexdigest() == meta[model]["offset_sha256"]


def test_errors_are_reported():
    from opendcvc_amd.entropy import EntropyCoder
    c = EntropyCoder()
    with pytest.raises(_lib.DcvcError):
        c.encode_y(np.zeros(4, np.int16), 3)          # unknown cdf group
    with pytest.raises(_lib.DcvcError):
        c.set_stream(b"\x00")                          # too short
    with pytest.raises(_lib.DcvcError):
        c.get_encoded_stream()                         # flush() not called


@pytest.mark.parametrize("two", [0, 1])
def test_entry_point_variants_agree(kat, two):
    """borrowed-buffer encode == copying encode; asynchronous decode_y + get == decode_and_get_y."""
    c = coder(kat, two)
    p = packed_of(kat)
    c.reset()
    c.encode_z(kat["z"], 0, 0, 6)
    c.encode_y(p, 0, borrowed=True)
    c.encode_y(p[:777], 0, borrowed=True)
    c.encode_y(p[:0], 0, borrowed=True)
    c.flush()
    s = c.get_encoded_stream()
    assert s == kat[f"stream_two{two}"].tobytes()
    c.set_stream(s)
    z = np.empty(kat["z"].size, np.int8)
    c.decode_z(z.size, 0, 0, 6)
    c.get_decoded(z)
    y = np.empty(kat["idx"].size, np.int8)
    c.decode_y(kat["idx"], 0)
    c.get_decoded(y)
    assert np.array_equal(y, kat["sym"].astype(np.int8))
    y2 = np.empty(777, np.int8)
    c.decode_and_get_y(np.ascontiguousarray(kat["idx"][:777]), 0, y2)
    assert np.array_equal(y2, kat["sym"][:777].astype(np.int8))
    with pytest.raises(_lib.DcvcError):
        c.decode_and_get_y(kat["idx"], 0, np.empty(3, np.int8))       # output too small


def test_real_1080p_frame_inputs(golden_dir):
    """The coder inputs of one 1080p P frame recorded on the MI355X box (tools/dump_coder_inputs.py):
    same stream as the oracle coder (itself pinned to the reference), and it decodes back."""
    from opendcvc_amd import entropy, weights
    d = np.load(os.path.join(golden_dir, "coder_inputs_1080p.npz"))
    sd = weights.make_state_dict("dmc", 1234)
    pre = "bit_estimator_z."
    params = {k[len(pre):]: torch.from_numpy(v) for k, v in sd.items() if k.startswith(pre)}
    ztab = entropy.factorized_cdf_tables(params, 72, 128)
    c = entropy.EntropyCoder()
    o = O.Coder()
    for t in (entropy.gaussian_cdf_tables(), ztab):
        c.add_cdf(*t)
        o.add_cdf(*t)
    c.set_use_two_entropy_coders(True)
    o.set_use_two(1)
    zg, zoff, zper = (int(v) for v in d["z_args"])
    yg = int(d["y_group"])
    c.reset()
    o.reset()
    c.encode_z(d["z"], zg, zoff, zper)
    o.encode_z(d["z"], zg, zoff, zper)
    packed = [d["packed0"], d["packed1"]]          # borrowed: must outlive get_encoded_stream()
    for p in packed:
        c.encode_y(p, yg, borrowed=True)
        o.encode_y(p[(p & 0xff) != 0xff], yg)
    c.flush()
    s = c.get_encoded_stream()
    assert s == d["stream"].tobytes() == o.flush()
    c.set_stream(s)
    z = np.empty(d["z"].size, np.int8)
    c.decode_z(z.size, zg, zoff, zper)
    c.get_decoded(z)
    assert np.array_equal(z, d["z"])
    for k in ("0", "1"):
        p, idx = d["packed" + k], d["index" + k]
        out = np.empty(idx.size, np.int8)
        c.decode_and_get_y(idx, yg, out)
        kept = idx != 0xff
        assert np.array_equal(out[kept], (p >> 8)[kept].astype(np.int8)) and not out[~kept].any()
