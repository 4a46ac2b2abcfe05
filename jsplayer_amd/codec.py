"""Host-side mirror of jsplayer's IVideoCodec plugin surface over the C ABI.

Same class names, method names, argument meaning and results as the reference
(IVideoCodec.hx:16-29; MSVideo1.hx:8,262; ScreenPressor.hx:19), so code written against the
Haxe interface reads the same here:

    dec = MSVideo1_16bit(320, 240)
    dec.Preinit(36)
    state = dec.DecompressI(frame_bytes, buf)          # DecoderState
    res = dec.DecompressP(frame_bytes, other_buf)      # PFrameResult(data_pnt, significant_changes)
    res.data_pnt is dec.PreviousFrame()                # identity, as Manager.hx:516 relies on

Frame buffers are caller-owned int32 arrays of at least width*height elements, 0x00RRGGBB,
bottom-up (Manager.hx:114-118): a CUDA/HIP `torch.int32` tensor (the frame stays in HBM) or a
C-contiguous `numpy.int32` array (host-pointer compatibility mode).
"""
from __future__ import annotations

import ctypes as C
import enum
import weakref
from dataclasses import dataclass
from typing import Any, Optional, Sequence

import numpy as np

from . import _native as N


class DecoderState(enum.IntEnum):
    """IVideoCodec.hx:5-9"""
    zero_state = 0
    in_progress = 1
    error_occured = 2


@dataclass
class PFrameResult:
    """IVideoCodec.hx:11-14"""
    data_pnt: Any
    significant_changes: bool


class CodecError(RuntimeError):
    pass


def _src_arg(src):
    """bytes / bytearray / memoryview / numpy uint8 -> (object to keep alive, pointer, length)"""
    if src is None:
        return None, None, 0
    if isinstance(src, (bytes, bytearray)):
        keep = bytes(src) if isinstance(src, bytearray) else src
        return keep, C.cast(C.c_char_p(keep), C.c_void_p), len(keep)
    arr = np.ascontiguousarray(np.frombuffer(src, dtype=np.uint8) if not isinstance(src, np.ndarray) else src,
                               dtype=np.uint8)
    return arr, C.c_void_p(arr.ctypes.data), arr.size


def _frame_ptr(buf, npixels: int) -> int:
    """Address of a caller-owned frame buffer (torch device tensor or numpy host array)."""
    if isinstance(buf, np.ndarray):
        if buf.dtype != np.int32 or not buf.flags["C_CONTIGUOUS"] or buf.size < npixels:
            raise CodecError("host frame buffer must be C-contiguous int32 with >= width*height elements")
        return buf.ctypes.data
    # torch tensor (duck-typed so importing this module does not import torch)
    if hasattr(buf, "data_ptr"):
        import torch
        if buf.dtype != torch.int32 or not buf.is_contiguous() or buf.numel() < npixels:
            raise CodecError("frame tensor must be contiguous int32 with >= width*height elements")
        return buf.data_ptr()
    raise CodecError(f"unsupported frame buffer type {type(buf)!r}")


class _NativeCodec:
    """Common part of the three codec classes: owns one jsp_codec handle."""

    _kind = 0

    def __init__(self, width: int, height: int, bpp: int = 0, palette: Optional[bytes] = None,
                 device: int = 0):
        self.X, self.Y = int(width), int(height)
        self._lib = N.lib()
        pal = bytes(palette) if palette is not None else None
        self._h = self._lib.jsp_codec_create(self._kind, self.X, self.Y, int(bpp), pal,
                                             len(pal) if pal else 0, int(device))
        if not self._h:
            raise CodecError(N.last_error())
        # address -> caller object, to hand identical objects back.  Weak: the codec keeps alive only what the
        # reference keeps alive — the current previous frame (`_prev`) — plus the frames of live staged batches
        # (held by the StagedBatch); a caller that allocates a fresh buffer per frame does not accumulate them here.
        self._bufs = weakref.WeakValueDictionary()
        self._prev = None
        self._inflight = {}   # ticket -> (src bytes, dst) of frames between *_async and wait()

    # -- IVideoCodec ------------------------------------------------------------------------
    def Preinit(self, insignificant_lines: int) -> None:
        if self._lib.jsp_preinit(self._h, int(insignificant_lines)) != 0:
            raise CodecError(N.last_error())

    def PreviousFrame(self):
        return self._prev

    def _track_prev(self):
        addr = self._lib.jsp_previous_frame(self._h)
        self._prev = self._bufs.get(addr) if addr else None

    def IsKeyFrame(self, data) -> bool:
        keep, p, n = _src_arg(data)
        return bool(self._lib.jsp_is_key_frame(self._h, p, n))

    def State(self) -> DecoderState:
        return DecoderState(self._lib.jsp_state(self._h))

    def ContinueI(self) -> DecoderState:
        return DecoderState(self._lib.jsp_continue_i(self._h))

    def DecompressI(self, src, dst) -> DecoderState:
        keep, p, n = _src_arg(src)
        addr = _frame_ptr(dst, self.X * self.Y)
        self._bufs[addr] = dst
        rc = self._lib.jsp_decompress_i(self._h, p, n, C.c_void_p(addr))
        self._track_prev()
        return DecoderState(rc)

    def DecompressP(self, src, dst) -> PFrameResult:
        keep, p, n = _src_arg(src)
        addr = _frame_ptr(dst, self.X * self.Y)
        self._bufs[addr] = dst
        out_ptr = C.c_void_p()
        signif = C.c_int(0)
        rc = self._lib.jsp_decompress_p(self._h, p, n, C.c_void_p(addr), C.byref(out_ptr), C.byref(signif))
        self._track_prev()
        if rc != 0:
            # the reference raises out of DecompressP here (e.g. TypeError on a null prevFrame)
            raise CodecError(N.last_error())
        data = self._prev if out_ptr.value else None   # *data_pnt is the previous frame after the call
        return PFrameResult(data, bool(signif.value))

    # -- asynchronous form of DecompressI / DecompressP (jsp_decompress_*_async ... jsp_wait) ------------------
    def DecompressI_async(self, src, dst) -> int:
        """Host stage now, uploads and kernels queued; returns a ticket for wait().  `src` and `dst` are kept alive here
        until then; device frame buffers only."""
        return self._submit(self._lib.jsp_decompress_i_async, src, dst, True)

    def DecompressP_async(self, src, dst) -> int:
        return self._submit(self._lib.jsp_decompress_p_async, src, dst, False)

    def _submit(self, fn, src, dst, key: bool) -> int:
        keep, p, n = _src_arg(src)
        addr = _frame_ptr(dst, self.X * self.Y)
        self._bufs[addr] = dst
        ticket = C.c_uint64(0)
        if fn(self._h, p, n, C.c_void_p(addr), C.byref(ticket)) != 0:
            raise CodecError(N.last_error())
        self._inflight[ticket.value] = (keep, dst, key)
        self._track_prev()
        return ticket.value

    def wait(self, ticket: int):
        """What the synchronous call would have returned for the frame submitted under `ticket` (tickets are waited for
        in submission order): a DecoderState for DecompressI_async, a PFrameResult for DecompressP_async (CodecError
        where DecompressP raises)."""
        if ticket not in self._inflight:
            raise CodecError(f"no frame in flight under ticket {ticket}")
        if ticket != next(iter(self._inflight)):
            # (checked here as well: jsp_wait refuses it without consuming anything, and the frame's bytes and buffer must stay
            # alive for as long as the native job holds them)
            raise CodecError("tickets are waited for in submission order")
        out_ptr, signif = C.c_void_p(), C.c_int(0)
        rc = self._lib.jsp_wait(self._h, ticket, C.byref(out_ptr), C.byref(signif))
        _, _, key = self._inflight.pop(ticket)   # jsp_wait consumes the oldest ticket whatever its result
        self._track_prev()
        if key:
            return DecoderState(rc)
        if rc != 0:
            raise CodecError(N.last_error())
        return PFrameResult(self._bufs.get(out_ptr.value) if out_ptr.value else None, bool(signif.value))

    def NeedsIndex(self) -> bool:
        return bool(self._lib.jsp_needs_index(self._h))

    def StopAndClean(self) -> None:
        if getattr(self, "_h", None):
            self._lib.jsp_codec_destroy(self._h)   # waits for whatever is still in flight
            self._h = None
        self._inflight = {}                        # ... only then are the frames' bytes and buffers let go
        self._bufs = weakref.WeakValueDictionary()
        self._prev = None

    # -- batched / resident-input extension ----------------------------------------------------
    def set_stream(self, hip_stream: Optional[int]) -> None:
        self._lib.jsp_set_stream(self._h, C.c_void_p(hip_stream) if hip_stream else None)

    def set_option(self, key: str, value: str) -> None:
        if self._lib.jsp_set_option(self._h, key.encode(), value.encode()) != 0:
            raise CodecError(f"option {key}={value} not accepted")

    def sync(self) -> None:
        if self._lib.jsp_sync(self._h) != 0:
            raise CodecError(N.last_error())

    def KeyFrameDiffers(self) -> Optional[bool]:
        """With option "key_frame_compare" = "<first row>": whether the last key frame (DecompressI, or the last one collected with
        wait) differs from the frame before it from that row on — the pixel loop of Manager.frames_differ_significantly
        (Manager.hx:413-419), worked out with the decode; None when there was nothing to compare with (jsp_key_frame_differs)."""
        v = int(self._lib.jsp_key_frame_differs(self._h))
        return None if v < 0 else bool(v)

    def prefetch(self, host) -> None:
        """jsp_prefetch: the next frames' bytes lie in `host` (a numpy uint8 array / memoryview over a stretch of the file, ideally
        in pinned memory: PinnedBytes.array slices) — the codec may take the whole range to the device in one copy; asynchronous
        frames whose `src` is a slice of it then queue no upload of their own.  `None` gives every range up.  The range must stay
        alive and unchanged while the codec keeps it (the 4 most recent ranges): the object is held here."""
        if host is None:
            self._lib.jsp_prefetch(self._h, None, 0)
            self._ranges = []
            return
        keep, p, n = _src_arg(host)
        if self._lib.jsp_prefetch(self._h, p, n) != 0:
            raise CodecError(N.last_error())
        self._ranges = (getattr(self, "_ranges", []) + [keep])[-4:]

    def counter(self, name: str) -> int:
        """How often this instance took one of its slow paths ("async_reruns", "lookback_fallbacks"): jsp_counter."""
        v = int(self._lib.jsp_counter(self._h, name.encode()))
        if v < 0:
            raise CodecError(f"no counter named {name}")
        return v

    def stage_batch(self, srcs: Sequence, dsts: Sequence, is_key: Optional[Sequence[bool]] = None,
                    reuse: Optional["StagedBatch"] = None) -> "StagedBatch":
        """jsp_stage_batch; with `reuse` (a batch of this codec whose decodes have finished) jsp_restage_batch: the batch
        object's buffers are taken over, `reuse` itself is returned, now holding this batch."""
        n = len(srcs)
        if len(dsts) != n:
            raise CodecError("srcs and dsts differ in length")
        keeps, ptrs, lens = [], (C.c_void_p * n)(), (C.c_size_t * n)()
        dptrs = (C.c_void_p * n)()
        for i, (s, d) in enumerate(zip(srcs, dsts)):
            keep, p, ln = _src_arg(s)
            keeps.append(keep)
            ptrs[i] = p.value if p is not None else None
            lens[i] = ln
            addr = _frame_ptr(d, self.X * self.Y)
            self._bufs[addr] = d
            dptrs[i] = addr
        keys = bytes(bytearray(1 if k else 0 for k in is_key)) if is_key is not None else None
        if reuse is not None and reuse._h:
            h = self._lib.jsp_restage_batch(self._h, reuse._h, n, ptrs, lens, keys, dptrs)
            if not h:
                raise CodecError(N.last_error())
            self._track_prev()
            reuse._h, reuse.n, reuse._dsts = h, n, list(dsts)
            return reuse
        h = self._lib.jsp_stage_batch(self._h, n, ptrs, lens, keys, dptrs)
        if not h:
            raise CodecError(N.last_error())
        self._track_prev()
        return StagedBatch(self, h, n, list(dsts))

    def DecompressI_batch(self, srcs: Sequence, dsts: Sequence) -> DecoderState:
        st = self.stage_batch(srcs, dsts)
        try:
            st.decode()
            self.sync()
            status, _, _ = st.results()
            bad = [s for s in status if s != 0]
            return DecoderState(bad[0] if bad else 0)
        finally:
            st.close()

    def __del__(self):
        try:
            self.StopAndClean()
        except Exception:
            pass


class StagedBatch:
    """A batch whose descriptor tables are resident in HBM (jsp_stage_batch)."""

    def __init__(self, codec: _NativeCodec, handle: int, n: int, dsts=()):
        self._codec, self._h, self.n = codec, handle, n
        self._dsts = dsts   # the kernels write here for as long as the batch can be decoded

    def decode(self) -> None:
        """Queue the reconstruction kernels (asynchronous on the codec's stream)."""
        if self._codec._lib.jsp_staged_decode(self._codec._h, self._h) != 0:
            raise CodecError(N.last_error())

    def info(self) -> dict:
        out = N.StagedInfo()
        self._codec._lib.jsp_staged_get_info(self._h, C.byref(out))
        return out.as_dict()

    def kernels(self) -> str:
        """Names of the kernels decode() launches, " + " separated (jsp_staged_kernels)."""
        return self._codec._lib.jsp_staged_kernels(self._h).decode()

    def results(self):
        st, ad, sg = (C.c_int * self.n)(), (C.c_int * self.n)(), (C.c_int * self.n)()
        self._codec._lib.jsp_staged_results(self._h, st, ad, sg)
        return list(st), list(ad), list(sg)

    def close(self) -> None:
        if self._h:
            self._codec._lib.jsp_staged_destroy(self._h)
            self._h = None
        self._dsts = ()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MSVideo1_16bit(_NativeCodec):
    """MSVideo1.hx:8-260 — `new MSVideo1_16bit(width, height)`"""
    _kind = N.JSP_CODEC_MSVIDEO1_16

    def __init__(self, width: int, height: int, device: int = 0):
        super().__init__(width, height, 16, None, device)


class MSVideo1_8bit(_NativeCodec):
    """MSVideo1.hx:262-429 — `new MSVideo1_8bit(width, height, palette)`; `palette` = the strf
    bytes after the BITMAPINFOHEADER (RGBQUADs), AVIParser.hx:79-85."""
    _kind = N.JSP_CODEC_MSVIDEO1_8

    def __init__(self, width: int, height: int, palette: bytes, device: int = 0):
        super().__init__(width, height, 8, palette, device)


class ScreenPressor(_NativeCodec):
    """ScreenPressor.hx:19-490 — `new ScreenPressor(width, height, bits_per_pixel)`"""
    _kind = N.JSP_CODEC_SCREENPRESSOR

    def __init__(self, width: int, height: int, bits_per_pixel: int, device: int = 0):
        super().__init__(width, height, bits_per_pixel, None, device)


class _DeviceView:
    """npixels int32 at a device address, for torch.as_tensor (the CUDA array interface; ROCm builds of torch honour it)."""

    def __init__(self, ptr: int, npixels: int):
        self.__cuda_array_interface__ = {"shape": (npixels,), "typestr": "<i4", "data": (ptr, False), "version": 2}


class FramePool:
    """The frame buffers a caller decodes into (Manager.hx:114-118, hx/FrameBuffer.hx): jsp_pool_create.  `frames` are torch int32
    tensors over the pool's buffers (valid until close()).  A pool of 32 frames or more is one allocation that the library PLACES —
    it measures what candidate allocations take from the decode kernels' store shape and keeps a fast one (include/jsplayer_amd.h);
    `store_rate` (GB/s, 0 for small pools) and `attempts` say what it found."""

    def __init__(self, width: int, height: int, count: int, device: int = 0):
        import torch
        self._lib = N.lib()
        self._h = self._lib.jsp_pool_create(device, width, height, count)
        if not self._h:
            raise CodecError(N.last_error())
        tried = C.c_int(0)
        self.store_rate = float(self._lib.jsp_pool_store_rate(self._h, C.byref(tried)))
        self.attempts = tried.value
        ms, held, limit = C.c_double(0), C.c_uint64(0), C.c_uint64(0)
        self._lib.jsp_pool_probe_info(self._h, C.byref(ms), C.byref(held), C.byref(limit))
        self.probe_ms, self.held_bytes, self.hold_limit = ms.value, held.value, limit.value   # what placing the pool cost
        rates = (C.c_double * 64)()
        n = self._lib.jsp_pool_probe_rates(self._h, rates, 64)
        self.tried_rates = [float(rates[i]) for i in range(max(0, min(n, 64)))]             # GB/s of every candidate measured, in order
        n = width * height
        self.frames = [torch.as_tensor(_DeviceView(int(self._lib.jsp_pool_buffer(self._h, i)), n), device=f"cuda:{device}") for i in range(count)]

    def close(self) -> None:
        if self._h:
            self.frames = []
            self._lib.jsp_pool_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HostBuffer:
    """Pinned host memory for compressed frames (jsp_host_alloc): uploads from it need no staging copy.  `.array` is a
    numpy uint8 view; frames handed to the *_async calls as slices of it are uploaded from where they are."""

    def __init__(self, nbytes: int):
        self._lib = N.lib()
        self._p = self._lib.jsp_host_alloc(max(int(nbytes), 1))
        if not self._p:
            raise CodecError("jsp_host_alloc failed")
        self.array = np.ctypeslib.as_array((C.c_uint8 * max(int(nbytes), 1)).from_address(self._p))

    def close(self) -> None:
        if self._p:
            self.array = None
            self._lib.jsp_host_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- the two Manager passes that follow the codec (Manager.hx:325-390, 413-419), on the GPU --------
DISPLAY_CANVAS, DISPLAY_CANVAS_RGB15, DISPLAY_SETPIXELS, DISPLAY_SETPIXELS_RGB15 = 0, 1, 2, 3


def display_convert(frame, out, width: int, height: int, mode: int = DISPLAY_CANVAS, flip_rows: bool = False,
                    stream: int = 0) -> None:
    """Manager.fill_bitmap_data on device tensors (int32, width*height)."""
    lib = N.lib()
    rc = lib.jsp_display_convert(C.c_void_p(_frame_ptr(frame, width * height)), C.c_void_p(_frame_ptr(out, width * height)),
                                 width, height, mode, 1 if flip_rows else 0, C.c_void_p(stream) if stream else None)
    if rc != 0:
        raise CodecError(N.last_error())


def frames_differ(a, b, first_pixel: int, npixels: int, stream: int = 0) -> bool:
    """The pixel compare of Manager.frames_differ_significantly on device tensors."""
    lib = N.lib()
    out = C.c_int(0)
    rc = lib.jsp_frames_differ(C.c_void_p(_frame_ptr(a, npixels)), C.c_void_p(_frame_ptr(b, npixels)), first_pixel, npixels,
                               C.byref(out), C.c_void_p(stream) if stream else None)
    if rc != 0:
        raise CodecError(N.last_error())
    return bool(out.value)
