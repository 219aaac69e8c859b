"""Bitstream container (opendcvc_amd/bitstream.py) against byte strings written by the reference's
stream_helper (tests/golden/container_kat.json): varints in every length class, SPS, I/P NAL units and a
whole multi-frame stream with SPS de-duplication; plus read-back."""
import hashlib
import io
import json
import os

import numpy as np
import pytest

from opendcvc_amd import bitstream as B
from opendcvc_amd.pipeline import FramePacket


@pytest.fixture(scope="module")
def kat(golden_dir):
    return json.load(open(os.path.join(golden_dir, "container_kat.json")))


def test_varints(kat):
    for c in kat["varint"]:
        f = io.BytesIO()
        assert B.write_uint_adaptive(f, c["value"]) == c["n"]
        assert f.getvalue().hex() == c["hex"]
        assert B.read_uint_adaptive(io.BytesIO(f.getvalue())) == c["value"]
    with pytest.raises(ValueError):
        B.write_uint_adaptive(io.BytesIO(), 1 << 30)
    with pytest.raises(EOFError):
        B.read_uint_adaptive(io.BytesIO(b"\xc0\x01"))


def test_sps(kat):
    for c in kat["sps"]:
        f = io.BytesIO()
        assert B.write_sps(f, c["sps"]) == c["n"]
        assert f.getvalue().hex() == c["hex"]
        r = io.BytesIO(f.getvalue())
        h = B.read_header(r)
        assert h["nal_type"] == B.NalType.NAL_SPS and h["sps_id"] == c["sps"]["sps_id"]
        assert B.read_sps_remaining(r, h["sps_id"]) == c["sps"]


def test_ip_units(kat):
    rng = np.random.default_rng(5)
    for c in kat["ip"]:
        payload = rng.integers(0, 256, c["payload_len"], dtype=np.uint8).tobytes()
        assert hashlib.sha256(payload).hexdigest() == c["payload_sha256"]
        f = io.BytesIO()
        assert B.write_ip(f, c["is_i"], c["sps_id"], c["qp"], payload) == c["n"]
        assert hashlib.sha256(f.getvalue()).hexdigest() == c["sha256"] and f.getvalue()[:8].hex() == c["hex_head"]
        r = io.BytesIO(f.getvalue())
        h = B.read_header(r)
        assert (h["nal_type"] == B.NalType.NAL_I) == c["is_i"] and h["sps_id"] == c["sps_id"]
        assert B.read_ip_remaining(r) == (c["qp"], payload)


def test_whole_stream_with_sps_dedup(kat):
    s = kat["stream"]
    rng = np.random.default_rng(5)
    for c in kat["ip"]:       # the generator drew the ip payloads first
        rng.integers(0, 256, c["payload_len"], dtype=np.uint8)
    f = io.BytesIO()
    w = B.StreamWriter(f)
    payloads = []
    for fr in s["frames"]:
        payload = rng.integers(0, 256, fr["payload_len"], dtype=np.uint8).tobytes()
        assert hashlib.sha256(payload).hexdigest() == fr["payload_sha256"]
        payloads.append(payload)
        w.write_frame(1080, 1920, True, FramePacket(fr["is_i"], fr["qp"], fr["use_ada_i"], payload))
    data = f.getvalue()
    assert len(data) == s["n"] and hashlib.sha256(data).hexdigest() == s["sha256"]
    r = B.StreamReader(io.BytesIO(data))
    for fr, payload in zip(s["frames"], payloads):
        sps, is_i, qp, got = r.read_frame()
        assert (sps["height"], sps["width"], sps["ec_part"], sps["use_ada_i"]) == (1080, 1920, 1, fr["use_ada_i"])
        assert is_i == fr["is_i"] and qp == fr["qp"] and got == payload


def test_sps_helper_limits():
    h = B.SPSHelper()
    for i in range(16):
        sid, new = h.get_sps_id({"height": 16 * (i + 1), "width": 16, "ec_part": 0, "use_ada_i": 0})
        assert sid == i and new
    assert h.get_sps_id({"height": 16, "width": 16, "ec_part": 0, "use_ada_i": 0}) == (0, False)
    with pytest.raises(ValueError):
        h.get_sps_id({"height": 999, "width": 16, "ec_part": 0, "use_ada_i": 0})
