"""Host side of the entropy model: CDF table construction (once per model) and the hand-off
between the GPU and the C++ rANS coder of libdcvc_amd.so.

Mirrors, with the same method names, the reference's
  EntropyCoder     src/models/entropy_models.py:11-81   (over MLCodec_extensions_cpp)
  GaussianEncoder  src/models/entropy_models.py:227-341 (scale table, index maths, tables)
  BitEstimator     src/models/entropy_models.py:129-224 (factorized prior of z)
Symbols travel as fixed-size arrays with a sentinel for skipped entries instead of the
reference's boolean-mask compaction (data-dependent size => device sync); the coder drops the
sentinels, so the byte stream is unchanged.
"""
import ctypes
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import _lib
from ._lib import DcvcError, check

SCALE_MIN, SCALE_MAX, SCALE_LEVELS = 0.11, 16.0, 128


def _ip(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def pmf_to_quantized_cdf(pmf, precision=16):
    pmf = np.ascontiguousarray(pmf, np.float32)
    out = np.zeros(pmf.size + 1, np.uint32)
    check(_lib.lib().dcvc_pmf_to_quantized_cdf(_ip(pmf), pmf.size, precision, _ip(out)), "pmf_to_quantized_cdf")
    return out


def _pmf_to_cdf(pmf, tail_mass, pmf_length, max_length):
    cdf = np.zeros((len(pmf_length), max_length + 2), np.int32)
    for i in range(len(pmf_length)):
        prob = np.concatenate([pmf[i, :pmf_length[i]], tail_mass[i]])
        c = pmf_to_quantized_cdf(prob)
        cdf[i, :c.size] = c.astype(np.int32)
    return cdf


def gaussian_cdf_tables():
    """128 quantised zero-mean Gaussian CDFs on the log-spaced scale table (entropy_models.py:244-283).
    Computed on the host CPU in fp32 so encoder and decoder always agree."""
    table = torch.exp(torch.linspace(math.log(SCALE_MIN), math.log(SCALE_MAX), SCALE_LEVELS))
    normal = torch.distributions.normal.Normal
    center = torch.full_like(table, 8.0)
    d = normal(0., table)
    for i in range(8, 1, -1):
        probs = d.cdf(torch.full_like(table, float(i)))
        center = torch.where(probs > 0.9999, torch.full_like(table, float(i)), center)
    center = center.int()
    length = 2 * center + 1
    max_length = int(length.max())
    samples = (torch.arange(max_length) - center[:, None]).float()
    d = normal(0., table[:, None].expand_as(samples))
    upper, lower = d.cdf(samples + 0.5), d.cdf(samples - 0.5)
    cdf = _pmf_to_cdf((upper - lower).numpy(), (2 * lower[:, :1]).numpy(), length.numpy(), max_length)
    return cdf, (length + 2).numpy().astype(np.int32), (-center).numpy().astype(np.int32)


def factorized_cdf_tables(params, qp_num, channel):
    """Per-(qp, channel) CDFs of the factorized z prior (entropy_models.py:152-205).
    params: dict 'f1.h' ... 'f4.b' -> float32 tensors [qp_num, channel, 1, 1] on the CPU."""
    def bitparm(x, f, final):
        x = x * F.softplus(params[f + ".h"]) + params[f + ".b"]
        return x if final else x + torch.tanh(x) * torch.tanh(params[f + ".a"])

    def cdf_of(x):
        for f in ("f1", "f2", "f3"):
            x = bitparm(x, f, False)
        return torch.sigmoid(bitparm(x, "f4", True))

    zero = torch.zeros((qp_num, channel, 1, 1))
    minima, maxima = zero + 8, zero + 8
    for i in range(8, 1, -1):
        minima = torch.where(cdf_of(zero - i) < 0.0001, zero + i, minima)
        maxima = torch.where(cdf_of(zero + i) > 0.9999, zero + i, maxima)
    minima, maxima = minima.int(), maxima.int()
    pmf_length = maxima + minima + 1
    max_length = int(pmf_length.max())
    samples = torch.arange(max_length)[None, None, None, :] + (zero - minima)
    lower, upper = cdf_of(samples - 0.5), cdf_of(samples + 0.5)
    pmf = (upper - lower)[:, :, 0, :]
    top = cdf_of(maxima.to(torch.float32))
    tail = lower[:, :, 0, :1] + (1.0 - top[:, :, 0, -1:])
    cdf = _pmf_to_cdf(pmf.reshape(-1, max_length).numpy(), tail.reshape(-1, 1).numpy(),
                      pmf_length.reshape(-1).numpy(), max_length)
    return cdf, (pmf_length.reshape(-1) + 2).numpy().astype(np.int32), (-minima).reshape(-1).numpy().astype(np.int32)


class PinnedBuffer:
    """hipHostMalloc'ed staging buffer viewed as a numpy array."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        self.ptr = _lib.lib().dcvc_host_alloc(self.nbytes)
        if not self.ptr:
            raise DcvcError("pinned host allocation failed")
        self.u8 = np.ctypeslib.as_array(ctypes.cast(self.ptr, ctypes.POINTER(ctypes.c_uint8)), (self.nbytes,))
        self._dptr = None

    @property
    def dptr(self):
        """device address of the buffer (kernels write the coder's input in place: no copy command)"""
        if self._dptr is None:
            self._dptr = _lib.lib().dcvc_host_device_ptr(ctypes.c_void_p(self.ptr))
            if not self._dptr:
                raise DcvcError("pinned buffer has no device address")
        return self._dptr

    def view(self, dtype, count):
        return self.u8[:count * np.dtype(dtype).itemsize].view(dtype)

    def __del__(self):
        try:
            _lib.lib().dcvc_host_free(ctypes.c_void_p(self.ptr))
        except Exception:
            pass


class EntropyCoder:
    """reference: EntropyCoder (entropy_models.py:11-81) + RansEncoder/RansDecoder (py_rans.cpp)."""

    def __init__(self):
        L = _lib.lib()
        self.enc = ctypes.c_void_p(L.dcvc_rans_enc_create())
        self.dec = ctypes.c_void_p(L.dcvc_rans_dec_create())
        if not self.enc or not self.dec:
            raise DcvcError("cannot create the rANS coder")
        self._pinned = {}

    def __del__(self):
        try:
            L = _lib.lib()
            L.dcvc_rans_enc_destroy(self.enc)
            L.dcvc_rans_dec_destroy(self.dec)
        except Exception:
            pass

    def pinned(self, key, nbytes):
        """Pinned staging buffer for (key, nbytes).  One buffer per distinct size, never replaced or freed while
        the coder lives: captured HIP graphs (models.GraphCache) keep the raw host pointer in their memcpy
        nodes, so a buffer handed out once must stay valid and keep its meaning for that (key, size)."""
        k = (key, int(nbytes))
        b = self._pinned.get(k)
        if b is None:
            b = self._pinned[k] = PinnedBuffer(nbytes)
        return b

    def adopt_pinned(self, other):
        """takes over another coder's staging buffers (CompressionModel.update() called again: graphs captured
        earlier may still reference them)"""
        if other is not None:
            self._pinned.update(other._pinned)

    def add_cdf(self, cdf, cdf_length, offset):
        L = _lib.lib()
        cdf = np.ascontiguousarray(cdf, np.int32)
        cdf_length = np.ascontiguousarray(cdf_length, np.int32)
        offset = np.ascontiguousarray(offset, np.int32)
        a = check(L.dcvc_rans_enc_add_cdf(self.enc, _ip(cdf), cdf.shape[0], cdf.shape[1], _ip(cdf_length), _ip(offset)), "add_cdf")
        b = check(L.dcvc_rans_dec_add_cdf(self.dec, _ip(cdf), cdf.shape[0], cdf.shape[1], _ip(cdf_length), _ip(offset)), "add_cdf")
        assert a == b
        return a

    def set_use_two_entropy_coders(self, two):
        L = _lib.lib()
        L.dcvc_rans_enc_set_use_two(self.enc, int(bool(two)))
        L.dcvc_rans_dec_set_use_two(self.dec, int(bool(two)))

    # ---- encoder
    def reset(self):
        check(_lib.lib().dcvc_rans_enc_reset(self.enc), "rans reset")

    def encode_y(self, symbols, cdf_group_index, borrowed=False):
        """symbols: host int16 array ((sym << 8) + index, index 0xFF = skipped).  borrowed=True skips the
        copy: the array must then stay untouched until get_encoded_stream() has returned."""
        symbols = np.ascontiguousarray(symbols, np.int16)
        f = _lib.lib().dcvc_rans_enc_encode_y_borrowed if borrowed else _lib.lib().dcvc_rans_enc_encode_y
        check(f(self.enc, _ip(symbols), symbols.size, cdf_group_index), "encode_y")

    def encode_z(self, symbols, cdf_group_index, start_offset, per_channel_size):
        symbols = np.ascontiguousarray(symbols, np.int8)
        check(_lib.lib().dcvc_rans_enc_encode_z(self.enc, _ip(symbols), symbols.size, cdf_group_index, start_offset,
                                                per_channel_size), "encode_z")

    def flush(self):
        check(_lib.lib().dcvc_rans_enc_flush(self.enc), "rans flush")

    def get_encoded_stream(self):
        p = ctypes.c_void_p()
        n = check(_lib.lib().dcvc_rans_enc_get_stream(self.enc, ctypes.byref(p)), "get_encoded_stream")
        return ctypes.string_at(p, n) if n else b""

    # ---- decoder
    def set_stream(self, stream):
        buf = np.frombuffer(stream, np.uint8)
        check(_lib.lib().dcvc_rans_dec_set_stream(self.dec, _ip(buf), buf.size), "set_stream")

    def decode_y(self, indexes, cdf_group_index):
        """indexes: host uint8 array, 0xFF = skipped (decodes to 0)."""
        indexes = np.ascontiguousarray(indexes, np.uint8)
        check(_lib.lib().dcvc_rans_dec_decode_y(self.dec, _ip(indexes), indexes.size, cdf_group_index), "decode_y")

    def decode_z(self, total_size, cdf_group_index, start_offset, per_channel_size):
        check(_lib.lib().dcvc_rans_dec_decode_z(self.dec, total_size, cdf_group_index, start_offset, per_channel_size), "decode_z")

    def get_decoded(self, out):
        """Blocks until the queued decode finished and copies the int8 symbols into `out` (host)."""
        n = check(_lib.lib().dcvc_rans_dec_get(self.dec, _ip(out), out.size), "get_decoded")
        return n

    def check_end(self):
        """After the last symbol of a frame: raises DcvcError if the payload was corrupt or truncated (the coder is not
        back in its initial state / has bytes left over).  No reference counterpart (it decodes garbage silently)."""
        check(_lib.lib().dcvc_rans_dec_check_end(self.dec), "corrupt or truncated frame payload")

    def decode_compact(self, indexes, count, cdf_group_index, out):
        """Synchronous: `indexes[:count]` are the KEPT table indexes only (compacted on the device, stream order); decodes one
        symbol each into out[:count]."""
        if indexes.dtype != np.uint8 or out.dtype != np.int8 or count < 0 or indexes.size < count or out.size < count or \
                not indexes.flags.c_contiguous or not out.flags.c_contiguous:
            raise DcvcError("decode_compact: need contiguous uint8 indexes and an int8 output of at least `count` entries")
        check(_lib.lib().dcvc_rans_dec_decode_compact(self.dec, _ip(indexes), int(count), cdf_group_index, _ip(out)),
              "decode_compact")
        return count

    def decode_and_get_y(self, indexes, cdf_group_index, out):
        """Synchronous: decodes straight from `indexes` into `out` (both host arrays of the same length)."""
        if indexes.dtype != np.uint8 or out.dtype != np.int8 or out.size < indexes.size or \
                not indexes.flags.c_contiguous or not out.flags.c_contiguous:
            raise DcvcError("decode_and_get_y: need contiguous uint8 indexes and an int8 output of the same length")
        check(_lib.lib().dcvc_rans_dec_decode_and_get_y(self.dec, _ip(indexes), indexes.size, cdf_group_index,
                                                        _ip(out)), "decode_and_get_y")
        return indexes.size
