"""ctypes binding of oracle/liboracle.so — the CPU restatement used as the CHECKER.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this.
The classes mirror the reference's IVideoCodec surface on numpy host buffers so that the parity
tests drive oracle and HIP path with identical calls.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_PATH = os.path.join(ROOT, "oracle", "liboracle.so")

_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(ORACLE_PATH)
        L.orc_msv1_create.restype = C.c_void_p
        L.orc_msv1_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int]
        L.orc_msv1_destroy.argtypes = [C.c_void_p]
        L.orc_msv1_preinit.argtypes = [C.c_void_p, C.c_int]
        L.orc_msv1_previous_frame.restype = C.c_void_p
        L.orc_msv1_previous_frame.argtypes = [C.c_void_p]
        L.orc_msv1_is_key_frame.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        L.orc_msv1_decompress_i.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p]
        L.orc_msv1_decompress_p.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p,
                                            C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
        _lib = L
    return _lib


class OracleAbort(RuntimeError):
    """The reference would raise out of the call (uncaught JS TypeError)."""


class OracleMSVideo1:
    def __init__(self, bits, width, height, palette=None):
        self.L = lib()
        self.X, self.Y = width, height
        pal = bytes(palette) if palette else None
        self.h = self.L.orc_msv1_create(bits, width, height, pal, len(pal) if pal else 0)
        assert self.h
        self._bufs = {}

    def Preinit(self, lines):
        self.L.orc_msv1_preinit(self.h, lines)

    def PreviousFrame(self):
        a = self.L.orc_msv1_previous_frame(self.h)
        return self._bufs.get(a) if a else None

    def IsKeyFrame(self, data):
        data = bytes(data)
        return bool(self.L.orc_msv1_is_key_frame(self.h, data, len(data)))

    def DecompressI(self, src, dst: np.ndarray):
        src = bytes(src)
        self._bufs[dst.ctypes.data] = dst
        rc = self.L.orc_msv1_decompress_i(self.h, src, len(src), C.c_void_p(dst.ctypes.data))
        return rc

    def DecompressP(self, src, dst: np.ndarray):
        src = bytes(src)
        self._bufs[dst.ctypes.data] = dst
        out, sg = C.c_void_p(), C.c_int()
        rc = self.L.orc_msv1_decompress_p(self.h, src, len(src), C.c_void_p(dst.ctypes.data),
                                          C.byref(out), C.byref(sg))
        if rc != 0:
            raise OracleAbort()
        return (self._bufs.get(out.value) if out.value else None), bool(sg.value)

    def close(self):
        if self.h:
            self.L.orc_msv1_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


def _sp_lib():
    L = lib()
    if not getattr(L, "_sp_ready", False):
        L.orc_sp_create.restype = C.c_void_p
        L.orc_sp_create.argtypes = [C.c_int, C.c_int, C.c_int]
        L.orc_sp_destroy.argtypes = [C.c_void_p]
        L.orc_sp_preinit.argtypes = [C.c_void_p, C.c_int]
        L.orc_sp_previous_frame.restype = C.c_void_p
        L.orc_sp_previous_frame.argtypes = [C.c_void_p]
        L.orc_sp_is_key_frame.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        L.orc_sp_decompress_i.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p]
        L.orc_sp_decompress_p.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p,
                                          C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
        L._sp_ready = True
    return L


class OracleScreenPressor:
    """oracle/screenpressor_oracle.cpp behind the IVideoCodec method names."""

    def __init__(self, width, height, bpp):
        self.L = _sp_lib()
        self.X, self.Y = width, height
        self.h = self.L.orc_sp_create(width, height, bpp)
        assert self.h
        self._bufs = {}

    def Preinit(self, lines):
        self.L.orc_sp_preinit(self.h, lines)

    def PreviousFrame(self):
        a = self.L.orc_sp_previous_frame(self.h)
        return self._bufs.get(a) if a else None

    def IsKeyFrame(self, data):
        data = bytes(data)
        return bool(self.L.orc_sp_is_key_frame(self.h, data, len(data)))

    def DecompressI(self, src, dst: np.ndarray):
        """0 zero_state, 2 error_occured, 3 the reference would raise / hang"""
        src = bytes(src)
        self._bufs[dst.ctypes.data] = dst
        return self.L.orc_sp_decompress_i(self.h, src, len(src), C.c_void_p(dst.ctypes.data))

    def DecompressP(self, src, dst: np.ndarray):
        src = bytes(src)
        self._bufs[dst.ctypes.data] = dst
        out, sg = C.c_void_p(), C.c_int()
        rc = self.L.orc_sp_decompress_p(self.h, src, len(src), C.c_void_p(dst.ctypes.data), C.byref(out), C.byref(sg))
        if rc != 0:
            raise OracleAbort()
        return (self._bufs.get(out.value) if out.value else None), bool(sg.value)

    def close(self):
        if self.h:
            self.L.orc_sp_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


def orc_display_convert(frame: np.ndarray, width: int, height: int, mode: int, flip_rows: bool) -> np.ndarray:
    """oracle/manager_oracle.cpp: Manager.fill_bitmap_data (Manager.hx:325-390) on a host frame."""
    L = lib()
    L.orc_display_convert.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    src = np.ascontiguousarray(frame, dtype=np.int32)
    out = np.empty(width * height, dtype=np.int32)
    L.orc_display_convert(C.c_void_p(src.ctypes.data), C.c_void_p(out.ctypes.data), width, height, mode, 1 if flip_rows else 0)
    return out


def orc_frames_differ(a: np.ndarray, b: np.ndarray, first_pixel: int, npixels: int) -> bool:
    """oracle/manager_oracle.cpp: the pixel loop of Manager.frames_differ_significantly (Manager.hx:413-419)."""
    L = lib()
    L.orc_frames_differ.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]
    a = np.ascontiguousarray(a, dtype=np.int32)
    b = np.ascontiguousarray(b, dtype=np.int32)
    return bool(L.orc_frames_differ(C.c_void_p(a.ctypes.data), C.c_void_p(b.ctypes.data), first_pixel, npixels))
