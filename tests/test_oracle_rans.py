"""The oracle's rANS / CDF restatement against the reference's golden vectors (bit-exact) and, when
oracle/_ref is present, against the reference coder itself on fresh random inputs."""
import glob
import os
import sys

import numpy as np
import pytest

import dcvc_oracle as O


def _coder(k, two):
    c = O.Coder()
    assert c.add_cdf(k["cdf"], k["sizes"], k["offsets"]) == 0
    c.set_use_two(two)
    return c


@pytest.fixture(scope="module")
def kat(golden_dir):
    return np.load(os.path.join(golden_dir, "rans_kat.npz"))


def _packed(k):
    return ((k["sym"].astype(np.int32) << 8) + k["idx"]).astype(np.int16)


@pytest.mark.parametrize("two", [0, 1])
def test_encode_matches_reference_stream(kat, two):
    c = _coder(kat, two)
    p = _packed(kat)
    c.reset()
    c.encode_z(kat["z"], 0, 0, 6)
    c.encode_y(p, 0)
    c.encode_y(p[:777], 0)
    c.encode_y(p[:0], 0)
    assert c.flush() == kat[f"stream_two{two}"].tobytes()
    small = ((np.clip(kat["sym"], -2, 2).astype(np.int32) << 8) + kat["idx"]).astype(np.int16)
    for m in (64, 201):
        c.reset()
        c.encode_y(small[:m], 0)
        assert c.flush() == kat[f"stream_two{two}_y{m}"].tobytes()


@pytest.mark.parametrize("two", [0, 1])
def test_decode_reference_stream(kat, two):
    c = _coder(kat, two)
    c.set_stream(kat[f"stream_two{two}"].tobytes())
    assert np.array_equal(c.decode_z(kat["z"].size, 0, 0, 6), kat["z"])
    assert np.array_equal(c.decode_y(kat["idx"], 0), kat["sym"].astype(np.int8))
    assert np.array_equal(c.decode_y(kat["idx"][:777], 0), kat["sym"][:777].astype(np.int8))


def test_pmf_to_quantized_cdf(kat):
    for p, c, n in zip(kat["pmf_in"], kat["pmf_out"], kat["pmf_len"]):
        got = O.pmf_to_quantized_cdf(p[:n], 16)
        assert np.array_equal(got.astype(np.int64), c[:n + 1])


def test_empty_flush():
    k = dict(cdf=np.array([[0, 30000, 65536, 0]], np.int32), sizes=np.array([3], np.int32), offsets=np.array([0], np.int32))
    c = _coder(k, 0)
    c.reset()
    assert c.flush() == b""


def test_gaussian_tables_match_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "gauss_cdf.npz"))
    cdf, length, offset = O.gaussian_tables()
    assert np.array_equal(cdf, g["cdf"])
    assert np.array_equal(length, g["length"])
    assert np.array_equal(offset, g["offset"])


def test_against_reference_build_random():
    """Fresh random inputs through oracle/_ref (the reference's coder compiled in place)."""
    here = os.path.dirname(os.path.abspath(O.__file__))
    if not glob.glob(os.path.join(here, "_ref", "MLCodec_extensions_cpp*.so")):
        pytest.skip("oracle/_ref not built")
    sys.path.insert(0, os.path.join(here, "_ref"))
    import MLCodec_extensions_cpp as R
    g = O.gaussian_tables()
    rng = np.random.default_rng(3)
    for two in (False, True):
        for n in (1000, 50001):
            idx = rng.integers(0, 128, n).astype(np.uint8)
            sigma = 0.11 * (16 / 0.11) ** (idx / 127.0)
            sym = np.clip(np.round(rng.standard_normal(n) * sigma), -128, 127).astype(np.int16)
            p = ((sym.astype(np.int32) << 8) + idx).astype(np.int16)
            enc = R.RansEncoder()
            enc.add_cdf(*g)
            enc.set_use_two_encoders(two)
            enc.reset()
            enc.encode_y(p, 0)
            enc.flush()
            ref = np.array(enc.get_encoded_stream()).tobytes()
            c = O.Coder()
            c.add_cdf(*g)
            c.set_use_two(two)
            c.reset()
            c.encode_y(p, 0)
            assert c.flush() == ref
            c.set_stream(ref)
            assert np.array_equal(c.decode_y(idx, 0), sym.astype(np.int8))
