"""Robustness of the host side against damaged input - all on the CPU.

The reference has no handling: RansDecoderLib reads renormalisation bytes through a raw pointer with no end check
(src/cpp/py_rans/rans.cpp:356-429) and stream_helper.py slices whatever is left of the file.  The drop-in must be better:
  * the host coder never reads outside the payload (AddressSanitizer + UBSan build, `make -C opendcvc_amd/csrc asan`,
    driver opendcvc_amd/csrc/rans_fuzz.cpp: round trips, every truncation length, bit flips, bad arguments);
  * a damaged payload is REPORTED: decode calls return -4 (DcvcError) when the stream runs out, and check_end() - called by
    DMC / DMCI.decompress after a frame's last symbol - rejects a coder that is not back in its initial state or has not
    consumed exactly its bytes;
  * a truncated or malformed container raises EOFError / ValueError."""
import io
import os
import subprocess

import numpy as np
import pytest

import dcvc_oracle as O
from opendcvc_amd import _lib
from opendcvc_amd import bitstream as B
from opendcvc_amd._lib import DcvcError
from opendcvc_amd.pipeline import FramePacket

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "opendcvc_amd", "csrc")


def test_host_coder_under_asan_ubsan():
    """builds the coder with -fsanitize=address,undefined -fno-sanitize-recover and runs the fuzz driver: any invalid read,
    signed overflow or bad shift aborts it; the driver itself checks the return codes and check_end()"""
    subprocess.run(["make", "-C", CSRC, "asan"], check=True, capture_output=True, timeout=600)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([os.path.join(REPO, "opendcvc_amd", "rans_fuzz_asan"), "48"], capture_output=True, text=True, env=env,
                       timeout=600)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    assert " 0 failures" in p.stdout and "truncations" in p.stdout


needs_lib = pytest.mark.skipif(not os.path.exists(_lib.LIB_PATH), reason="libdcvc_amd.so not built")


def _frame(two, n=20011, seed=3):
    """one frame's worth of symbols the way the models hand them over: z + two halves of y with sentinels"""
    from opendcvc_amd.entropy import EntropyCoder
    rng = np.random.default_rng(seed)
    g = O.gaussian_tables()
    idx = rng.integers(0, 128, (2, n)).astype(np.uint8)
    sigma = 0.11 * (16 / 0.11) ** (idx / 127.0)
    sym = np.clip(np.round(rng.standard_normal((2, n)) * sigma), -128, 127).astype(np.int16)
    keep = rng.random((2, n)) < 0.3
    idx = np.where(keep, idx, 0xFF).astype(np.uint8)
    z = rng.integers(-3, 4, 640).astype(np.int8)
    c = EntropyCoder()
    c.add_cdf(*g)
    c.set_use_two_entropy_coders(two)
    c.reset()
    c.encode_y(((sym[0].astype(np.int32) << 8) + idx[0]).astype(np.int16), 0)      # (z through the Gaussian tables too:
    c.encode_z(z, 0, 60, 64)                                                        #  any table group will do here)
    c.encode_y(((sym[1].astype(np.int32) << 8) + idx[1]).astype(np.int16), 0)
    c.flush()
    return c, c.get_encoded_stream(), idx, np.where(keep, sym, 0).astype(np.int8), z


def _decode(c, stream, idx, nz):
    c.set_stream(stream)
    out0 = np.empty(idx.shape[1], np.int8)
    c.decode_and_get_y(idx[0], 0, out0)
    z = np.empty(nz, np.int8)
    c.decode_z(nz, 0, 60, 64)
    c.get_decoded(z)
    out1 = np.empty(idx.shape[1], np.int8)
    c.decode_and_get_y(idx[1], 0, out1)
    c.check_end()
    return out0, z, out1


@needs_lib
@pytest.mark.parametrize("two", [0, 1])
def test_intact_payload_passes_check_end(two):
    c, stream, idx, sym, z = _frame(two)
    o0, oz, o1 = _decode(c, stream, idx, z.size)
    assert np.array_equal(o0, sym[0]) and np.array_equal(o1, sym[1]) and np.array_equal(oz, z)


@needs_lib
@pytest.mark.parametrize("two", [0, 1])
def test_truncated_payload_is_reported(two):
    c, stream, idx, sym, z = _frame(two)
    for cut in (0, 3, 4, 5, len(stream) // 3, len(stream) // 2, len(stream) - 9, len(stream) - 1):
        with pytest.raises(DcvcError):
            _decode(c, stream[:cut], idx, z.size)
    with pytest.raises(DcvcError):
        _decode(c, stream + b"\x00", idx, z.size)           # one byte too many is not this frame's payload either
    _decode(c, stream, idx, z.size)                          # the coder is reusable after a refused stream


@needs_lib
@pytest.mark.parametrize("two", [0, 1])
def test_bit_flipped_payload_is_reported(two):
    """no escapes in this frame's symbols (|sym| stays inside the tables' support for most sigma) is not guaranteed: a flip
    in an escaped value's verbatim bits is a valid stream of another frame (see rans_fuzz.cpp) - so: either DcvcError, or
    the only differences are at escaped positions"""
    c, stream, idx, sym, z = _frame(two)
    g_cdf, g_len, g_off = O.gaussian_tables()
    rng = np.random.default_rng(11)
    reported = 0
    for _ in range(60):
        bad = bytearray(stream)
        pos = int(rng.integers(0, len(bad)))
        bad[pos] ^= 1 << int(rng.integers(0, 8))
        try:
            o0, oz, o1 = _decode(c, bytes(bad), idx, z.size)
        except DcvcError as e:
            assert "corrupt or truncated" in str(e)
            reported += 1
            continue
        for o, s, ix in ((o0, sym[0], idx[0]), (o1, sym[1], idx[1])):
            diff = np.nonzero(o != s)[0]
            t = ix[diff].astype(np.int64)
            v = s[diff].astype(np.int64) - g_off[t]
            assert np.all((v < 0) | (v >= g_len[t] - 2)), "a table-coded symbol changed and the payload was accepted"
    assert reported >= 50


@needs_lib
def test_decoder_argument_errors_raise():
    from opendcvc_amd.entropy import EntropyCoder
    c = EntropyCoder()
    c.add_cdf(*O.gaussian_tables())
    with pytest.raises(DcvcError):
        c.set_stream(b"\x01\x02\x03")
    c.set_stream(b"\x00" * 16)
    with pytest.raises(DcvcError):
        c.decode_and_get_y(np.full(8, 200, np.uint8), 0, np.empty(8, np.int8))      # table 200 of 128
    with pytest.raises(DcvcError):
        c.decode_z(10, 3, 0, 1)                                                      # unknown group
    with pytest.raises(DcvcError):
        c.decode_z(1000, 0, 120, 1)                                                  # channels beyond the group


# ----------------------------------------------------------------------------------------------- container

def _container():
    f = io.BytesIO()
    w = B.StreamWriter(f)
    w.write_frame(1080, 1920, True, FramePacket(True, 21, 0, bytes(range(200))))
    w.write_frame(1080, 1920, True, FramePacket(False, 29, 0, bytes(300)))
    w.write_frame(1080, 1920, True, FramePacket(False, 25, 1, bytes(20000)))         # 4-byte length, new SPS
    return f.getvalue()


def test_truncated_container_raises_never_returns_a_short_frame():
    data = _container()
    r = B.StreamReader(io.BytesIO(data))
    frames = [r.read_frame() for _ in range(3)]
    assert [len(f[3]) for f in frames] == [200, 300, 20000] and [f[2] for f in frames] == [21, 29, 25]
    with pytest.raises(EOFError):
        r.read_frame()                                        # clean end of file
    for cut in list(range(0, 12)) + [150, 207, 208, 209, 211, 400, 520, 530, len(data) - 1]:
        r = B.StreamReader(io.BytesIO(data[:cut]))
        got = 0
        with pytest.raises(EOFError):
            for _ in range(3):
                sps, is_i, qp, payload = r.read_frame()
                assert len(payload) == (200, 300, 20000)[got]      # whatever does come back is a whole frame
                got += 1


def test_malformed_container_raises():
    data = bytearray(_container())
    bad = bytes([0x70]) + bytes(data[1:])                     # NAL type 7 does not exist
    with pytest.raises(ValueError):
        B.StreamReader(io.BytesIO(bad)).read_frame()
    with pytest.raises(ValueError):                           # a frame that names an SPS the stream never carried
        B.StreamReader(io.BytesIO(bytes([(int(B.NalType.NAL_P) << 4) | 5, 30, 4, 1, 2, 3, 4]))).read_frame()
    # a length field larger than what is left
    with pytest.raises(EOFError):
        sps = bytes(data[:6])
        B.StreamReader(io.BytesIO(sps + bytes([(int(B.NalType.NAL_I) << 4) | 0, 30, 0xc0, 0x10, 0, 0]) + bytes(100))).read_frame()
