"""ctypes binding of libdcvc_amd.so (include/dcvc_amd.h).

The HIP library is the product: there is NO Python/torch fallback for any operator.  If the
shared object is missing or a call fails, an exception is raised (DcvcError) - never a silent
CPU path.
"""
import ctypes
import os
import subprocess
from ctypes import (POINTER, c_char_p, c_float, c_int, c_int8, c_int16, c_int32, c_int64, c_size_t,
                    c_uint8, c_uint32, c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
# DCVC_AMD_DIAG=1 selects the developer build with in-kernel diagnostics (make -C opendcvc_amd/csrc diag)
# DCVC_AMD_LIB=<file name in this directory> selects an experimental build (kernel A/B measurements)
LIB_PATH = os.path.join(_HERE, os.environ.get("DCVC_AMD_LIB") or
                        ("libdcvc_amd_diag.so" if os.environ.get("DCVC_AMD_DIAG") else "libdcvc_amd.so"))

F16, F32 = 0, 1
EPI_BIAS, EPI_BIAS_QUANT, EPI_SHUFFLE2, EPI_WSILU = 0, 1, 2, 3


class DcvcError(RuntimeError):
    pass


def build(force=False):
    """Compiles opendcvc_amd/csrc into opendcvc_amd/libdcvc_amd.so with hipcc for gfx950."""
    csrc = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-C", csrc, "clean"])
    subprocess.check_call(["make", "-C", csrc, "-j8"])
    return LIB_PATH


_P = c_void_p
_I = c_int
_L = c_int64
_F = c_float

# name -> (restype, argtypes); mirrors include/dcvc_amd.h one to one
_SIGS = {
    "dcvc_abi_version": (_I, []),
    "dcvc_last_error": (c_char_p, []),
    "dcvc_device_count": (_I, []),
    "dcvc_dcb_create": (_I, [_I, _I, _I, _I] + [_P] * 12 + [POINTER(_P)]),
    "dcvc_dcb_destroy": (None, [_P]),
    "dcvc_dcb_scratch_bytes": (c_size_t, [_P, _I, _I]),
    "dcvc_dcb_forward": (_I, [_P, _P, _L, _I, _P, _L, _I, _I, _I, _P, _P, _L, _P, _P]),
    "dcvc_dcb_forward_chained": (_I, [_P, _P, _L, _I, _P, _L, _I, _I, _I, _P, _P, _L, _P, _P, _I, _I, _P]),
    "dcvc_dcb_forward_then_conv": (_I, [_P, _P, _L, _I, _P, _L, _I, _I, _I, _P, _P, _I, _I, _P, _P, _P, _L]),
    "dcvc_dcb_profile": (_I, [_P, _P, _L, _I, _I, _I, _P, _L, _P, _P, _I, POINTER(c_float), POINTER(c_float)]),
    "dcvc_dcb_profile_tail": (_I, [_P, _P, _L, _I, _I, _I, _P, _L, _P, _P, _I, POINTER(c_float)]),
    "dcvc_conv_create": (_I, [_I, _I, _I, _I, _I, _I, _I, _I, _P, _P, POINTER(_P)]),
    "dcvc_conv_destroy": (None, [_P]),
    "dcvc_conv_forward": (_I, [_P, _P, _L, _I, _P, _L, _I, _I, _I, _P, _P, _L, _P]),
    "dcvc_conv_forward_scaled": (_I, [_P, _P, _L, _I, _P, _L, _I, _I, _I, _P, _P, _P, _L, _P]),
    "dcvc_unshuffle8": (_I, [_I, _P, _I, _I, _I, _P, _L, _P]),
    "dcvc_shuffle8_clamp": (_I, [_I, _P, _L, _P, _I, _I, _I, _I, _P, _P]),
    "dcvc_yuv420_to_frame": (_I, [_I, _P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "dcvc_frame_to_yuv420": (_I, [_I, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "dcvc_rgb_to_frame": (_I, [_I, _P, _I, _I, _I, _I, _P, _P]),
    "dcvc_frame_to_rgb": (_I, [_I, _P, _I, _I, _I, _I, _P, _P]),
    "dcvc_replicate_pad_hwc": (_I, [_I, _P, _L, _I, _I, _I, _I, _I, _P, _L, _P]),
    "dcvc_scale_channels": (_I, [_I, _P, _L, _P, _L, _I, _P, _L, _P]),
    "dcvc_copy_channels": (_I, [_I, _P, _L, _L, _I, _P, _L, _P]),
    "dcvc_crop_hwc": (_I, [_I, _P, _L, _I, _I, _I, _I, _P, _L, _P]),
    "dcvc_round_z": (_I, [_I, _P, _L, _I, _I, _I, _P, _P]),
    "dcvc_z_from_int8": (_I, [_I, _P, _I, _I, _I, _P, _L, _P]),
    "dcvc_prior_enc_step": (_I, [_I, _I, _I, _I, _P, _L, _P, _L, _P, _L, _P, _L, _I, _I, _I, _F, _P, _L, _P, _L, _P, _P]),
    "dcvc_prior_dec_index": (_I, [_I, _I, _I, _P, _L, _I, _I, _I, _F, _P, _P]),
    "dcvc_prior_dec_restore": (_I, [_I, _I, _I, _P, _P, _L, _I, _I, _I, _P, _L, _P, _L, _P]),
    "dcvc_prior_dec_compact_ws_bytes": (_L, [_I, _I, _I, _I]),
    "dcvc_prior_dec_index_compact": (_I, [_I, _I, _I, _P, _L, _I, _I, _I, _F, _P, _P, _P, _P, _P]),
    "dcvc_prior_dec_restore_compact": (_I, [_I, _I, _I, _P, _P, _P, _P, _L, _I, _I, _I, _P, _L, _P, _L, _P]),
    "dcvc_prior_finish": (_I, [_I, _I, _P, _L, _P, _L, _I, _I, _I, _P]),
    "dcvc_op_process_with_mask": (_I, [_I, _P, _P, _P, _P, _F, _P, _P, _P, _P, _L, _P]),
    "dcvc_op_combine_for_reading_2x": (_I, [_I, _P, _P, _P, _L, _P]),
    "dcvc_op_restore_y_2x": (_I, [_I, _P, _P, _P, _P, _L, _P]),
    "dcvc_op_restore_y_4x": (_I, [_I, _P, _P, _P, _P, _L, _P]),
    "dcvc_op_build_index_dec": (_I, [_I, _P, _P, _P, _F, _F, _F, _F, _F, _L, _P]),
    "dcvc_op_build_index_enc": (_I, [_I, _P, _P, _P, _P, _F, _F, _F, _F, _F, _L, _P]),
    "dcvc_op_round_and_to_int8": (_I, [_I, _P, _P, _L, _P]),
    "dcvc_op_clamp_reciprocal_with_quant": (_I, [_I, _P, _P, _F, _P, _L, _P]),
    "dcvc_op_add_and_multiply": (_I, [_I, _P, _P, _P, _L, _P]),
    "dcvc_op_bias_quant": (_I, [_I, _P, _P, _P, _I, _L, _P]),
    "dcvc_op_bias_pixel_shuffle_8": (_I, [_I, _P, _P, _P, _I, _I, _I, _I, _P]),
    "dcvc_op_replicate_pad": (_I, [_I, _P, _I, _I, _I, _I, _I, _P, _P]),
    "dcvc_op_bias_wsilu_depthwise_conv2d": (_I, [_I, _P, _P, _P, _I, _I, _I, _P, _P]),
    "dcvc_nchw_to_hwc": (_I, [_I, _P, _I, _L, _P, _L, _P]),
    "dcvc_hwc_to_nchw": (_I, [_I, _P, _L, _I, _L, _P, _P]),
    "dcvc_rans_enc_create": (_P, []),
    "dcvc_rans_enc_destroy": (None, [_P]),
    "dcvc_rans_enc_add_cdf": (_I, [_P, _P, _I, _I, _P, _P]),
    "dcvc_rans_enc_empty_cdf": (_I, [_P]),
    "dcvc_rans_enc_set_use_two": (None, [_P, _I]),
    "dcvc_rans_enc_reset": (_I, [_P]),
    "dcvc_rans_enc_encode_y": (_I, [_P, _P, _L, _I]),
    "dcvc_rans_enc_encode_y_borrowed": (_I, [_P, _P, _L, _I]),
    "dcvc_rans_enc_encode_z": (_I, [_P, _P, _L, _I, _I, _I]),
    "dcvc_rans_enc_flush": (_I, [_P]),
    "dcvc_rans_enc_get_stream": (_L, [_P, POINTER(_P)]),
    "dcvc_rans_dec_create": (_P, []),
    "dcvc_rans_dec_destroy": (None, [_P]),
    "dcvc_rans_dec_add_cdf": (_I, [_P, _P, _I, _I, _P, _P]),
    "dcvc_rans_dec_empty_cdf": (_I, [_P]),
    "dcvc_rans_dec_set_use_two": (None, [_P, _I]),
    "dcvc_rans_dec_set_stream": (_I, [_P, _P, _L]),
    "dcvc_rans_dec_decode_y": (_I, [_P, _P, _L, _I]),
    "dcvc_rans_dec_decode_z": (_I, [_P, _L, _I, _I, _I]),
    "dcvc_rans_dec_get": (_L, [_P, _P, _L]),
    "dcvc_rans_dec_decode_and_get_y": (_I, [_P, _P, _L, _I, _P]),
    "dcvc_rans_dec_decode_compact": (_I, [_P, _P, _L, _I, _P]),
    "dcvc_rans_dec_check_end": (_I, [_P]),
    "dcvc_pmf_to_quantized_cdf": (_I, [_P, _I, _I, _P]),
    "dcvc_host_alloc": (_P, [c_size_t]),
    "dcvc_host_free": (None, [_P]),
    "dcvc_host_device_ptr": (_P, [_P]),
    "dcvc_compact_symbols": (_I, [_P, _I, _I, _P, _P, _P, _P]),
    "dcvc_copy_f32": (_I, [_P, _P, _I, _P]),
    "dcvc_memcpy_d2h": (_I, [_P, _P, c_size_t, _P]),
    "dcvc_memcpy_h2d": (_I, [_P, _P, c_size_t, _P]),
    "dcvc_stream_sync": (_I, [_P]),
}

EXPORTS = tuple(_SIGS.keys())
_lib = None


def lib():
    """Returns the loaded library; raises DcvcError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DcvcError(
                f"{LIB_PATH} not found: the HIP extension has not been built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C opendcvc_amd/csrc`). "
                "There is no CPU fallback.")
        try:
            L = ctypes.CDLL(LIB_PATH)
        except OSError as e:
            raise DcvcError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in _SIGS.items():
            try:
                fn = getattr(L, name)
            except AttributeError as e:
                raise DcvcError(f"{LIB_PATH} does not export {name}") from e
            fn.restype = res
            fn.argtypes = args
        if L.dcvc_abi_version() != 1:
            raise DcvcError("libdcvc_amd.so ABI version mismatch")
        _lib = L
    return _lib


def check(rc, what=""):
    if rc is not None and rc < 0:
        msg = lib().dcvc_last_error()
        raise DcvcError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")
    return rc


def require_gpu():
    n = lib().dcvc_device_count()
    if n <= 0:
        raise DcvcError("no HIP device visible: the DCVC-RT hot path needs an MI355X (no CPU fallback)")
    return n
