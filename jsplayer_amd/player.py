"""Manager.worker-equivalent decode loop: the direct caller of the plugin surface,
kept to what touches the codec — construction from VideoInfo (Manager.hx:103-142), the frame-buffer
pool that never hands out the buffer holding the previous frame (:424-443,470-477), the
DecompressI / DecompressP protocol with its identity test (:499-524) and
`frames_differ_significantly` for key frames (:392-421).  Timers, seeking, bitmaps and audio of the
reference's Manager are not rebuilt.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence

import numpy as np

from .avi import CODEC_MSVC16, CODEC_MSVC8, CODEC_SCREENPRESSOR, VideoInfo

INSIGNIFICANT_LINES = 36  # Manager.hx:61
NUM_BUFFERS = 8           # Main.hx:148 (the pool holds NUM_BUFFERS + 1 frames, Manager.hx:114-118)


@dataclass
class DecodedFrame:
    index: int
    key: bool
    buffer_index: int            # pool slot holding the picture shown for this frame
    significant_changes: Optional[bool]
    state: int = 0               # DecoderState of a key frame


def make_decoder(vi: VideoInfo, classes) -> object:
    """`classes` = (MSVideo1_16bit, MSVideo1_8bit, ScreenPressor) — the HIP codecs or the oracle's."""
    m16, m8, sp = classes
    if vi.codec == CODEC_SCREENPRESSOR:
        return sp(vi.X, vi.Y, vi.bpp)
    if vi.codec == CODEC_MSVC16:
        return m16(vi.X, vi.Y)
    if vi.codec == CODEC_MSVC8:
        return m8(vi.X, vi.Y, vi.palette or b"")
    raise ValueError(vi.codec)


def _differ(a, b, start: int) -> bool:
    if isinstance(a, np.ndarray):
        return bool(np.any(a[start:] != b[start:]))
    from .codec import frames_differ   # device tensors: HIP reduction, the frames stay in HBM
    return frames_differ(a, b, start, a.numel())


class Manager:
    """Feeds compressed frames to an IVideoCodec in order, with the reference's buffer discipline."""

    def __init__(self, vi: VideoInfo, decoder, alloc: Callable[[int], object], num_buffers: int = NUM_BUFFERS):
        self.vi, self.decoder = vi, decoder
        self.buffers = [alloc(vi.X * vi.Y) for _ in range(num_buffers + 1)]
        self.holds: List[Optional[range]] = [None] * len(self.buffers)  # frames each slot currently shows
        self.decoder.Preinit(INSIGNIFICANT_LINES)
        # HIP codecs compare a key frame with the frame before it while they decode it (option "key_frame_compare"): no pass of the
        # Manager's own over the two frames (Manager.hx:413-419)
        self._fused_compare = hasattr(self.decoder, "KeyFrameDiffers")
        if self._fused_compare:
            self.decoder.set_option("key_frame_compare", str(INSIGNIFICANT_LINES))
        self.next_frame_to_decode = 0
        self.frame_of_interest = 0
        self.log: List[DecodedFrame] = []

    def _slot_of(self, buf) -> int:
        for i, b in enumerate(self.buffers):
            if b is buf:
                return i
        return -1

    def _get_free_buffer(self, prev_idx: int) -> int:  # Manager.hx:424-443
        oldest, oldest_first = -1, 1 << 30
        for i, h in enumerate(self.holds):
            if i == prev_idx:
                continue
            if h is None:
                return i
            if h.stop - 1 < self.frame_of_interest and h.start < oldest_first:
                oldest, oldest_first = i, h.start
        if oldest >= 0:
            self.holds[oldest] = None
        return oldest

    def worker(self, frame: bytes, index: int, prev_key_bytes: Optional[bytes], key: Optional[bool] = None) -> DecodedFrame:
        """One decode tick for compressed frame `index` (Manager.hx:454-539).  `key` = the flag the loader
        attached to the frame (an AVI index's, DataLoader.hx:373-401); None = scan the bytes."""
        dec = self.decoder
        if key is None:
            key = index == 0 or dec.IsKeyFrame(frame)      # DataLoaderAVISeq.hx:45
        prev = dec.PreviousFrame()
        prev_idx = self._slot_of(prev) if prev is not None else -1
        self.frame_of_interest = index                       # sequential playback keeps up with decode
        free = self._get_free_buffer(prev_idx)
        assert free >= 0
        new = self.buffers[free]
        if key:
            state = int(dec.DecompressI(frame, new))
            sig: Optional[bool] = None
            if state == 0:
                self.holds[free] = range(index, index + 1)
                # frames_differ_significantly, Manager.hx:392-421
                if index == 0:
                    sig = True
                elif prev_key_bytes is not None:
                    sig = prev_key_bytes != frame
                elif prev is None:
                    sig = True
                elif self._fused_compare:
                    sig = dec.KeyFrameDiffers()
                    sig = True if sig is None else sig
                else:
                    sig = _differ(new, prev, INSIGNIFICANT_LINES * self.vi.X)
            out = DecodedFrame(index, True, free, sig, state)
        else:
            res = dec.DecompressP(frame, new)
            shown = free
            if res.data_pnt is not None:
                if res.data_pnt is prev:                     # "no changes": the old slot keeps showing
                    h = self.holds[prev_idx]
                    self.holds[prev_idx] = range(h.start, index + 1) if h else range(index, index + 1)
                    shown = prev_idx
                else:
                    self.holds[free] = range(index, index + 1)
            out = DecodedFrame(index, False, shown, res.significant_changes)
        self.log.append(out)
        self.next_frame_to_decode = index + 1
        return out

    def play(self, frames: Sequence[bytes], on_frame: Optional[Callable[[DecodedFrame, object], None]] = None,
             key_flags: Optional[Sequence[bool]] = None):
        prev_key = None
        for i, f in enumerate(frames):
            was_key = bool(key_flags[i]) if key_flags is not None else (i == 0 or self.decoder.IsKeyFrame(f))
            d = self.worker(f, i, prev_key if was_key and i > 0 and self._last_was_key else None, was_key)
            self._last_was_key = was_key
            prev_key = f if was_key else prev_key
            if on_frame:
                on_frame(d, self.buffers[d.buffer_index])
        return self.log

    _last_was_key = False

    def play_pipelined(self, frames: Sequence[bytes], depth: int = 4,
                       on_frame: Optional[Callable[[DecodedFrame, object], None]] = None,
                       key_flags: Optional[Sequence[bool]] = None, prefetch_bytes: int = 0):
        """play() through the asynchronous calls (DecompressI_async / DecompressP_async / wait): up to `depth` frames are in
        flight, the host stage of frame n+1 runs while the uploads and kernels of frame n do.  The log is the one play()
        writes.  HIP codecs only; the pool must hold `depth` more buffers than play() needs (Manager(..., num_buffers =
        NUM_BUFFERS + depth)): a frame in flight keeps its destination AND the buffer it was decoded against."""
        from collections import deque
        from .codec import CodecError
        dec = self.decoder
        dec.set_option("async_depth", str(depth))
        # prefetch_bytes > 0 (MSVideo1): the frames are laid out in ONE pinned arena, as the file holds them (a chunk header in front of
        # each), and the arena goes to the device in ranges of about that size — the range after the current one ahead of the frames
        # being submitted (decoder.prefetch / jsp_prefetch); the frames then queue no upload of their own.  examples/jsp_play --prefetch.
        arena, spans, ahead = None, [], deque()
        if prefetch_bytes > 0 and hasattr(dec, "prefetch") and len(frames):
            from .codec import HostBuffer
            import numpy as np
            arena = HostBuffer(sum(len(f) + 8 + (len(f) & 1) for f in frames) + 16)
            pos = 0
            for f in frames:
                pos += 8
                arena.array[pos:pos + len(f)] = np.frombuffer(bytes(f), dtype=np.uint8)
                spans.append((pos, pos + len(f)))
                pos += len(f) + (len(f) & 1)
            frames = [arena.array[a:b] for a, b in spans]

        def fetch(i0):
            lo, j, hi = spans[i0][0] - 8, i0, spans[i0][1]
            while j < len(spans) and (j == i0 or spans[j][1] - lo <= prefetch_bytes):
                hi = spans[j][1]
                j += 1
            dec.prefetch(arena.array[lo:hi])
            ahead.append((i0, j))
        flying: deque = deque()      # (ticket, index, key, slot, prev_slot, prev buffer, compare bytes?, frame bytes)

        def collect():
            ticket, index, key, slot, prev_slot, prev, cmp_bytes, blob, prev_blob = flying.popleft()
            got = dec.wait(ticket)
            if key:
                state, sig = int(got), None
                if state == 0:
                    if index == 0:
                        sig = True
                    elif cmp_bytes:
                        sig = not (len(prev_blob) == len(blob) and bytes(prev_blob) == bytes(blob))
                    elif prev is None:
                        sig = True
                    elif self._fused_compare:
                        sig = dec.KeyFrameDiffers()      # (of the key frame just collected)
                        sig = True if sig is None else sig
                    else:
                        sig = _differ(self.buffers[slot], prev, INSIGNIFICANT_LINES * self.vi.X)
                out = DecodedFrame(index, True, slot, sig, state)
            else:
                shown = slot
                if got.data_pnt is not None and got.data_pnt is prev and prev_slot >= 0:
                    shown = prev_slot
                out = DecodedFrame(index, False, shown, got.significant_changes)
            self.log.append(out)
            if on_frame:
                on_frame(out, self.buffers[out.buffer_index])

        prev_key, last_was_key = None, False
        for i, f in enumerate(frames):
            if len(flying) == depth:
                collect()
            if arena is not None:
                while ahead and not (ahead[0][0] <= i < ahead[0][1]):
                    ahead.popleft()
                if not ahead:
                    fetch(i)
                if len(ahead) < 2 and ahead[-1][1] < len(spans):
                    fetch(ahead[-1][1])
            key = bool(key_flags[i]) if key_flags is not None else (i == 0 or dec.IsKeyFrame(f))
            prev = dec.PreviousFrame()                      # as of the last SUBMITTED frame
            prev_slot = self._slot_of(prev) if prev is not None else -1
            busy = {prev_slot} | {fl[3] for fl in flying} | {fl[4] for fl in flying}
            # what may still be looked at: everything from the oldest frame in flight on
            horizon = flying[0][1] - 1 if flying else i
            slot, oldest, oldest_first = -1, -1, 1 << 30
            for k, h in enumerate(self.holds):
                if k in busy:
                    continue
                if h is None:
                    slot = k
                    break
                if h.stop - 1 < horizon and h.start < oldest_first:
                    oldest, oldest_first = k, h.start
            if slot < 0:
                slot = oldest
                if slot >= 0:
                    self.holds[slot] = None
            if slot < 0:
                raise CodecError("no free frame buffer: the pool needs NUM_BUFFERS + depth + 1 buffers")
            dst = self.buffers[slot]
            ticket = dec.DecompressI_async(f, dst) if key else dec.DecompressP_async(f, dst)
            now = dec.PreviousFrame()                        # adoption is decided by the host stage: known at submission
            if now is dst:
                self.holds[slot] = range(i, i + 1)
            elif now is not None and now is prev and prev_slot >= 0:
                h = self.holds[prev_slot]
                self.holds[prev_slot] = range(h.start, i + 1) if h else range(i, i + 1)
            flying.append((ticket, i, key, slot, prev_slot, prev, key and last_was_key and i > 0, f, prev_key))
            last_was_key = key
            prev_key = f if key else prev_key
            self.next_frame_to_decode = i + 1
        while flying:
            collect()
        if arena is not None:
            dec.prefetch(None)                              # (the arena goes away: the codec must not find frames in it any more)
            arena.close()
        return self.log
