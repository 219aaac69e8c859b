"""Drop-in for the reference's host-coder module ``MLCodec_extensions_cpp`` (seam 3) on top
of the C ABI of libdcvc_amd.so (``dcvc_rans_*``, csrc/rans_host.cpp).

Same names, same argument meaning and the same byte streams as the pybind11 module the reference builds from
``src/cpp/py_rans/py_rans.cpp:366-393``::

    RansEncoder:  encode_y(int16[]), encode_z(int8[], group, start_offset, per_channel_size), flush(),
                  get_encoded_stream() -> uint8[], reset(), add_cdf(int32[n, L], int32[n], int32[n]) -> group,
                  empty_cdf_buffer(), set_use_two_encoders(bool), get_use_two_encoders()
    RansDecoder:  set_stream(uint8[]), decode_y(uint8[], group), decode_and_get_y(uint8[], group) -> int8[],
                  decode_z(total, group, start_offset, per_channel_size), get_decoded_tensor() -> int8[],
                  add_cdf(...), empty_cdf_buffer(), set_use_two_decoders(bool), get_use_two_decoders()
    pmf_to_quantized_cdf(list[float], precision) -> list[int]

so that the reference's own ``src/models/entropy_models.py`` (``from MLCodec_extensions_cpp import RansEncoder,
RansDecoder``) runs on this coder unchanged once the module is registered::

    import sys, opendcvc_amd.mlcodec_shim as shim
    sys.modules["MLCodec_extensions_cpp"] = shim        # or: shim.install()

Like the reference module, inputs are copied on entry, work is queued to the coder's worker threads and the ``get_*``
calls block.  Errors of the C ABI surface as ``DcvcError`` (the reference aborts or asserts).
"""
import ctypes
import sys

import numpy as np

from . import _lib
from ._lib import DcvcError, check

__all__ = ["RansEncoder", "RansDecoder", "pmf_to_quantized_cdf", "install"]


def _ip(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _tables(cdfs, cdfs_sizes, offsets):
    cdfs = np.ascontiguousarray(np.asarray(cdfs), np.int32)
    sizes = np.ascontiguousarray(np.asarray(cdfs_sizes), np.int32).reshape(-1)
    offs = np.ascontiguousarray(np.asarray(offsets), np.int32).reshape(-1)
    if cdfs.ndim != 2 or cdfs.shape[0] != sizes.size or offs.size != sizes.size:
        raise DcvcError("add_cdf: expected cdfs [n, L], cdfs_sizes [n], offsets [n]")
    return cdfs, sizes, offs


class RansEncoder:
    """py_rans.cpp:14-165"""

    def __init__(self):
        self._h = ctypes.c_void_p(_lib.lib().dcvc_rans_enc_create())
        if not self._h:
            raise DcvcError("cannot create the rANS encoder")
        self._two = False

    def __del__(self):
        try:
            _lib.lib().dcvc_rans_enc_destroy(self._h)
        except Exception:
            pass

    def encode_y(self, symbols, cdf_group_index):
        s = np.ascontiguousarray(np.asarray(symbols), np.int16).reshape(-1)
        check(_lib.lib().dcvc_rans_enc_encode_y(self._h, _ip(s), s.size, int(cdf_group_index)), "encode_y")

    def encode_z(self, symbols, cdf_group_index, start_offset, per_channel_size):
        s = np.ascontiguousarray(np.asarray(symbols), np.int8).reshape(-1)
        check(_lib.lib().dcvc_rans_enc_encode_z(self._h, _ip(s), s.size, int(cdf_group_index), int(start_offset),
                                                int(per_channel_size)), "encode_z")

    def flush(self):
        check(_lib.lib().dcvc_rans_enc_flush(self._h), "flush")

    def get_encoded_stream(self):
        p = ctypes.c_void_p()
        n = check(_lib.lib().dcvc_rans_enc_get_stream(self._h, ctypes.byref(p)), "get_encoded_stream")
        return np.frombuffer(ctypes.string_at(p, n), np.uint8).copy() if n else np.zeros(0, np.uint8)

    def reset(self):
        check(_lib.lib().dcvc_rans_enc_reset(self._h), "reset")

    def add_cdf(self, cdfs, cdfs_sizes, offsets):
        c, s, o = _tables(cdfs, cdfs_sizes, offsets)
        return check(_lib.lib().dcvc_rans_enc_add_cdf(self._h, _ip(c), c.shape[0], c.shape[1], _ip(s), _ip(o)), "add_cdf")

    def empty_cdf_buffer(self):
        check(_lib.lib().dcvc_rans_enc_empty_cdf(self._h), "empty_cdf_buffer")

    def set_use_two_encoders(self, b):
        self._two = bool(b)
        _lib.lib().dcvc_rans_enc_set_use_two(self._h, int(self._two))

    def get_use_two_encoders(self):
        return self._two


class RansDecoder:
    """py_rans.cpp:167-305"""

    def __init__(self):
        self._h = ctypes.c_void_p(_lib.lib().dcvc_rans_dec_create())
        if not self._h:
            raise DcvcError("cannot create the rANS decoder")
        self._two = False
        self._pending = 0          # size of the result of the last decode_* call

    def __del__(self):
        try:
            _lib.lib().dcvc_rans_dec_destroy(self._h)
        except Exception:
            pass

    def set_stream(self, encoded):
        e = np.ascontiguousarray(np.asarray(encoded), np.uint8).reshape(-1)
        check(_lib.lib().dcvc_rans_dec_set_stream(self._h, _ip(e), e.size), "set_stream")

    def decode_y(self, indexes, cdf_group_index):
        i = np.ascontiguousarray(np.asarray(indexes), np.uint8).reshape(-1)
        check(_lib.lib().dcvc_rans_dec_decode_y(self._h, _ip(i), i.size, int(cdf_group_index)), "decode_y")
        self._pending = i.size

    def decode_and_get_y(self, indexes, cdf_group_index):
        self.decode_y(indexes, cdf_group_index)
        return self.get_decoded_tensor()

    def decode_z(self, total_size, cdf_group_index, start_offset, per_channel_size):
        check(_lib.lib().dcvc_rans_dec_decode_z(self._h, int(total_size), int(cdf_group_index), int(start_offset),
                                                int(per_channel_size)), "decode_z")
        self._pending = int(total_size)

    def get_decoded_tensor(self):
        out = np.empty(self._pending, np.int8)
        n = check(_lib.lib().dcvc_rans_dec_get(self._h, _ip(out), out.size), "get_decoded_tensor")
        return out[:n]

    def add_cdf(self, cdfs, cdfs_sizes, offsets):
        c, s, o = _tables(cdfs, cdfs_sizes, offsets)
        return check(_lib.lib().dcvc_rans_dec_add_cdf(self._h, _ip(c), c.shape[0], c.shape[1], _ip(s), _ip(o)), "add_cdf")

    def empty_cdf_buffer(self):
        check(_lib.lib().dcvc_rans_dec_empty_cdf(self._h), "empty_cdf_buffer")

    def set_use_two_decoders(self, b):
        self._two = bool(b)
        _lib.lib().dcvc_rans_dec_set_use_two(self._h, int(self._two))

    def get_use_two_decoders(self):
        return self._two


def pmf_to_quantized_cdf(pmf, precision):
    """py_rans.cpp:307-364: list of floats -> list of precision-bit cumulative counts (len + 1 entries)"""
    p = np.ascontiguousarray(np.asarray(pmf, np.float32)).reshape(-1)
    out = np.zeros(p.size + 1, np.uint32)
    check(_lib.lib().dcvc_pmf_to_quantized_cdf(_ip(p), p.size, int(precision), _ip(out)), "pmf_to_quantized_cdf")
    return [int(v) for v in out]


def install(name="MLCodec_extensions_cpp"):
    """Registers this module under the reference's extension name, so `from MLCodec_extensions_cpp import ...`
    inside the reference's entropy_models.py resolves to the coder of libdcvc_amd.so."""
    sys.modules[name] = sys.modules[__name__]
    return sys.modules[name]
