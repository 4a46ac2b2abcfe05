"""Plain synchronous RIFF/AVI reader + writer for the input side of the decode path
(SURVEY.md §8f-2).  The reference parses AVI incrementally with parser combinators while bytes
trickle in over XHR (AVIParser.hx:142-171); only the facts that reach the codec are kept here:

* `avih` (AVIParser.hx:42-62): µs per frame (0 -> 66666), total frames, width, height;
* `strh` `vids` (:162-163): fourcc at +4, frame count at +32;
* `strf` (:64-88): biBitCount at +14, biCompression at +16 when the strh fourcc is 0, palette =
  bytes 40.. when biBitCount == 8; fourcc MSVC / msvc / CRAM / 0 selects MSVideo1 (8- or 16-bit),
  anything else ScreenPressor;
* `00dc` / `00db` chunks inside `LIST movi` (also inside `LIST rec `) are the compressed frames; the
  blob handed to the codec is the chunk size ROUNDED UP TO EVEN (ParserUtils.hx:24-27), i.e. it
  includes the pad byte;
* key frames: frame 0, or `IsKeyFrame(bytes)` (DataLoaderAVISeq.hx:45).
"""
from __future__ import annotations

import struct
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

CODEC_SCREENPRESSOR, CODEC_MSVC16, CODEC_MSVC8 = "screenpressor", "msvc16", "msvc8"
_MSVC_FOURCCS = {b"MSVC", b"msvc", b"CRAM", b"\0\0\0\0"}


@dataclass
class VideoInfo:
    """VideoData.hx:82-91"""
    X: int
    Y: int
    bpp: int
    fps: float
    nframes: int
    codec: str
    palette: Optional[bytes]
    riff_size: int


def _chunk(tag: bytes, payload: bytes) -> bytes:
    return tag + struct.pack("<I", len(payload)) + payload + (b"\0" if len(payload) & 1 else b"")


def write_avi(width: int, height: int, frames: Sequence[bytes], fourcc: bytes = b"CRAM", bpp: int = 16,
              fps: float = 15.0, palette: Optional[bytes] = None, key_flags: Optional[Sequence[bool]] = None) -> bytes:
    """A minimal single-video-stream AVI 1.0 file: hdrl(avih, strl(strh, strf)), movi(00dc...), idx1."""
    usec = int(round(1e6 / fps))
    n = len(frames)
    avih = struct.pack("<14I", usec, 0, 0, 0x10, n, 0, 1, 0, width, height, 0, 0, 0, 0)
    strh = b"vids" + fourcc + struct.pack("<IHHIIIIIIII4H", 0, 0, 0, 0, 1, int(round(fps)), 0, n, 0, 0xFFFFFFFF, 0,
                                           0, 0, width, height)
    pal = palette or b""
    strf = struct.pack("<IiiHH4sIiiII", 40, width, height, 1, bpp, fourcc, 0, 0, 0, len(pal) // 4 if bpp == 8 else 0, 0) + \
        (pal if bpp == 8 else b"")
    strl = b"strl" + _chunk(b"strh", strh) + _chunk(b"strf", strf)
    hdrl = b"hdrl" + _chunk(b"avih", avih) + _chunk(b"LIST", strl)
    movi = b"movi"
    index = b""
    for i, f in enumerate(frames):
        key = key_flags[i] if key_flags is not None else i == 0
        index += b"00dc" + struct.pack("<III", 0x10 if key else 0, len(movi), len(f))
        movi += _chunk(b"00dc", bytes(f))
    body = b"AVI " + _chunk(b"LIST", hdrl) + _chunk(b"LIST", movi) + _chunk(b"idx1", index)
    return b"RIFF" + struct.pack("<I", len(body)) + body


def read_avi(data: bytes) -> Tuple[VideoInfo, List[bytes]]:
    """Returns the video description and the frame blobs exactly as the reference hands them to
    `IVideoCodec` (chunk payload plus its pad byte when the size is odd)."""
    if data[:4] != b"RIFF" or data[8:12] != b"AVI ":
        raise ValueError("not a RIFF/AVI file")
    riff_size = struct.unpack_from("<I", data, 4)[0]
    info = dict(X=0, Y=0, bpp=32, fps=15.0, nframes=0, codec=CODEC_SCREENPRESSOR, palette=None)
    frames: List[bytes] = []
    state = dict(fourcc=None, is_video=False, have_video=False)

    def walk(lo: int, hi: int, in_movi: bool):
        pos = lo
        while pos + 8 <= hi:
            tag = data[pos:pos + 4]
            size = struct.unpack_from("<I", data, pos + 4)[0]
            body = pos + 8
            padded = (size + 1) & ~1
            if tag == b"LIST":
                kind = data[body:body + 4]
                walk(body + 4, min(body + size, hi), in_movi or kind == b"movi")
            elif tag == b"avih":
                usec, _, _, _, total = struct.unpack_from("<5I", data, body)
                w, h = struct.unpack_from("<2I", data, body + 32)
                info.update(X=w, Y=h, nframes=total, fps=1e6 / (usec if usec else 66666))
            elif tag == b"strh":
                state["is_video"] = data[body:body + 4] == b"vids" and not state["have_video"]
                if state["is_video"]:
                    state["fourcc"] = data[body + 4:body + 8]
                    info["nframes"] = struct.unpack_from("<I", data, body + 32)[0]
            elif tag == b"strf" and state["is_video"]:
                blob = data[body:body + padded]
                bits = struct.unpack_from("<H", blob, 14)[0]
                info["bpp"] = bits
                fourcc = state["fourcc"]
                if fourcc == b"\0\0\0\0":
                    fourcc = blob[16:20]
                if fourcc in _MSVC_FOURCCS:
                    info["codec"] = CODEC_MSVC8 if bits == 8 else CODEC_MSVC16
                if bits == 8 and len(blob) > 40:
                    info["palette"] = blob[40:]
                state["is_video"], state["have_video"] = False, True
            elif in_movi and tag in (b"00dc", b"00db"):
                frames.append(data[body:body + padded])
            pos = body + padded

    walk(12, min(len(data), 8 + riff_size), False)
    vi = VideoInfo(riff_size=riff_size, **info)
    return vi, frames
