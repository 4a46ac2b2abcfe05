"""ctypes binding of libjsplayer_amd.so (the C ABI in include/jsplayer_amd.h).

There is no CPU fallback: if the shared library is missing this module raises, loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libjsplayer_amd.so")

JSP_CODEC_MSVIDEO1_16 = 1
JSP_CODEC_MSVIDEO1_8 = 2
JSP_CODEC_SCREENPRESSOR = 3

JSP_ZERO_STATE = 0
JSP_IN_PROGRESS = 1
JSP_ERROR_OCCURED = 2


class StagedInfo(C.Structure):
    _fields_ = [
        ("frames", C.c_uint64),
        ("pixels", C.c_uint64),
        ("stream_bytes", C.c_uint64),
        ("descriptor_bytes", C.c_uint64),
        ("units_coded", C.c_uint64),
        ("units_copied", C.c_uint64),
        ("runs", C.c_uint64),
        ("algorithmic_bytes", C.c_uint64),
        ("kernel_launches", C.c_uint64),
        ("host_stage_ms", C.c_double),
        ("h2d_ms", C.c_double),
        ("device_parse_ms", C.c_double),
        ("moved_bytes", C.c_uint64),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


# name -> (restype, argtypes); kept in one table so tests can check every declared symbol
SIGNATURES = {
    "jsp_codec_create": (C.c_void_p, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int]),
    "jsp_codec_destroy": (None, [C.c_void_p]),
    "jsp_preinit": (C.c_int, [C.c_void_p, C.c_int]),
    "jsp_previous_frame": (C.c_void_p, [C.c_void_p]),
    "jsp_is_key_frame": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "jsp_state": (C.c_int, [C.c_void_p]),
    "jsp_continue_i": (C.c_int, [C.c_void_p]),
    "jsp_decompress_i": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "jsp_decompress_p": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                   C.POINTER(C.c_void_p), C.POINTER(C.c_int)]),
    "jsp_needs_index": (C.c_int, [C.c_void_p]),
    "jsp_last_error": (C.c_char_p, []),
    "jsp_pool_create": (C.c_void_p, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "jsp_pool_store_rate": (C.c_double, [C.c_void_p, C.POINTER(C.c_int)]),
    "jsp_pool_probe_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "jsp_pool_probe_rates": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_int]),
    "jsp_pool_buffer": (C.c_void_p, [C.c_void_p, C.c_int]),
    "jsp_pool_count": (C.c_int, [C.c_void_p]),
    "jsp_pool_destroy": (None, [C.c_void_p]),
    "jsp_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "jsp_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "jsp_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "jsp_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p]),
    "jsp_sync": (C.c_int, [C.c_void_p]),
    "jsp_counter": (C.c_longlong, [C.c_void_p, C.c_char_p]),
    "jsp_key_frame_differs": (C.c_int, [C.c_void_p]),
    "jsp_prefetch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "jsp_measure_h2d": (C.c_int, [C.c_int, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "jsp_device_count": (C.c_int, []),
    "jsp_assign_stream": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.c_int]),
    "jsp_reduce_counters": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]),
    "jsp_shard_last_error": (C.c_char_p, []),
    "jsp_decompress_i_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_uint64)]),
    "jsp_decompress_p_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_uint64)]),
    "jsp_wait": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]),
    "jsp_host_alloc": (C.c_void_p, [C.c_size_t]),
    "jsp_host_free": (None, [C.c_void_p]),
    "jsp_decompress_i_batch": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                         C.POINTER(C.c_void_p)]),
    "jsp_stage_batch": (C.c_void_p, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                     C.c_void_p, C.POINTER(C.c_void_p)]),
    "jsp_restage_batch": (C.c_void_p, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                       C.c_void_p, C.POINTER(C.c_void_p)]),
    "jsp_staged_decode": (C.c_int, [C.c_void_p, C.c_void_p]),
    "jsp_staged_destroy": (None, [C.c_void_p]),
    "jsp_staged_get_info": (C.c_int, [C.c_void_p, C.POINTER(StagedInfo)]),
    "jsp_staged_kernels": (C.c_char_p, [C.c_void_p]),
    "jsp_staged_results": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "jsp_display_convert": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "jsp_measure_fill": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_double), C.c_void_p]),
    "jsp_frames_differ": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.POINTER(C.c_int), C.c_void_p]),
    "jsp_version": (C.c_char_p, []),
}

_lib = None


def lib() -> C.CDLL:
    """Load (once) and return the native library; raise if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C jsplayer_amd/csrc). jsplayer_amd has no CPU fallback."
        )
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64/libhsa-runtime64, so
    # when torch is importable load it FIRST and let this library bind to the runtime torch brought
    # (loading /opt/rocm's copy first and torch's second leaves the second one without a device).
    try:
        import torch  # noqa: F401
    except Exception:  # torch is plumbing, not a dependency of the C ABI
        pass
    handle = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(handle, name)  # AttributeError if the ABI and the binding drift apart
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = handle
    return _lib


def last_error() -> str:
    return lib().jsp_last_error().decode("utf-8", "replace")
