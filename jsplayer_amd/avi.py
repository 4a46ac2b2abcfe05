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
* key frames: frame 0, or `IsKeyFrame(bytes)` (DataLoaderAVISeq.hx:45) — unless the file carries an
  index, whose flags then replace the scan (DataLoader.hx:373-401): `idx1` (AVI 1.0, flag bit 4,
  DataLoaderAVIIndexed.hx:276-350) or OpenDML `indx` super index -> `ix00` standard indexes (bit 31 of
  the size = NOT a key frame, VideoData.hx:26-39, DataLoader.hx:321-371).  `read_avi_indexed` walks the
  index instead of the `movi` list, which is also how the reference seeks.
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
              fps: float = 15.0, palette: Optional[bytes] = None, key_flags: Optional[Sequence[bool]] = None,
              opendml_frames_per_ix: int = 0) -> bytes:
    """A minimal single-video-stream AVI file: hdrl(avih, strl(strh, strf)), movi(00dc...), idx1.
    With `opendml_frames_per_ix` > 0 the index is OpenDML instead: an `indx` super index in the strl
    and one `ix00` chunk after every that many frames inside `movi` (no idx1)."""
    if opendml_frames_per_ix > 0:
        return _write_avi_opendml(width, height, frames, fourcc, bpp, fps, palette, key_flags, opendml_frames_per_ix)
    hdrl = _headers(width, height, len(frames), fourcc, bpp, fps, palette)
    movi = b"movi"
    index = b""
    for i, f in enumerate(frames):
        key = key_flags[i] if key_flags is not None else i == 0
        index += b"00dc" + struct.pack("<III", 0x10 if key else 0, len(movi), len(f))
        movi += _chunk(b"00dc", bytes(f))
    body = b"AVI " + _chunk(b"LIST", hdrl) + _chunk(b"LIST", movi) + _chunk(b"idx1", index)
    return b"RIFF" + struct.pack("<I", len(body)) + body


def _headers(width, height, n, fourcc, bpp, fps, palette, indx: bytes = b"") -> bytes:
    usec = int(round(1e6 / fps))
    avih = struct.pack("<14I", usec, 0, 0, 0x10, n, 0, 1, 0, width, height, 0, 0, 0, 0)
    strh = b"vids" + fourcc + struct.pack("<IHHIIIIIIII4H", 0, 0, 0, 0, 1, int(round(fps)), 0, n, 0, 0xFFFFFFFF, 0,
                                           0, 0, width, height)
    pal = palette or b""
    strf = struct.pack("<IiiHH4sIiiII", 40, width, height, 1, bpp, fourcc, 0, 0, 0, len(pal) // 4 if bpp == 8 else 0, 0) + \
        (pal if bpp == 8 else b"")
    strl = b"strl" + _chunk(b"strh", strh) + _chunk(b"strf", strf) + (_chunk(b"indx", indx) if indx else b"")
    return b"hdrl" + _chunk(b"avih", avih) + _chunk(b"LIST", strl)


def _write_avi_opendml(width, height, frames, fourcc, bpp, fps, palette, key_flags, per_ix) -> bytes:
    n = len(frames)
    nseg = max(1, (n + per_ix - 1) // per_ix)
    # super index: wLongsPerEntry=4, subtype 0, type 0 (index of indexes), nEntries, chunk id, 12 reserved
    indx_len = 24 + 16 * nseg
    hdrl = _headers(width, height, n, fourcc, bpp, fps, palette, b"\0" * indx_len)
    movi_tag_pos = 12 + 8 + len(hdrl) + 8          # RIFF hdr, LIST hdrl, LIST hdr of movi -> position of 'movi'
    movi = b"movi"
    supers = []
    for sidx in range(nseg):
        lo, hi = sidx * per_ix, min(n, (sidx + 1) * per_ix)
        entries = b""
        for i in range(lo, hi):
            f = bytes(frames[i])
            key = key_flags[i] if key_flags is not None else i == 0
            data_pos = movi_tag_pos + len(movi) + 8            # absolute position of the payload
            entries += struct.pack("<II", data_pos, len(f) | (0 if key else 0x80000000))
            movi += _chunk(b"00dc", f)
        # standard index chunk: wLongsPerEntry=2, subtype 0, type 1, nEntries, chunk id, qwBaseOffset=0, reserved
        ix = struct.pack("<HBBI4sQI", 2, 0, 1, hi - lo, b"00dc", 0, 0) + entries
        supers.append((movi_tag_pos + len(movi), 8 + len(ix), hi - lo))
        movi += _chunk(b"ix00", ix)
    indx = struct.pack("<HBBI4s12x", 4, 0, 0, nseg, b"00dc") + b"".join(struct.pack("<QII", o, sz, d) for o, sz, d in supers)
    assert len(indx) == indx_len
    hdrl = _headers(width, height, n, fourcc, bpp, fps, palette, indx)
    body = b"AVI " + _chunk(b"LIST", hdrl) + _chunk(b"LIST", movi)
    return b"RIFF" + struct.pack("<I", len(body)) + body


@dataclass
class IndexEntry:
    """VideoData.hx:25-39 after `base_offset` is applied: where the chunk HEADER is, payload size, key flag."""
    offset: int
    size: int
    key: bool


def read_index(data: bytes) -> Optional[List[IndexEntry]]:
    """The video stream's frame index, OpenDML first (as DataLoaderAVIIndexed.hx:149-158 prefers it),
    else idx1; None when the file has neither."""
    if data[:4] != b"RIFF" or data[8:12] != b"AVI ":
        raise ValueError("not a RIFF/AVI file")
    end = min(len(data), 8 + struct.unpack_from("<I", data, 4)[0])
    found = dict(indx=None, movi_size_pos=None, idx1=None)

    def walk(lo, hi, depth):
        pos = lo
        while pos + 8 <= hi:
            tag, size = data[pos:pos + 4], struct.unpack_from("<I", data, pos + 4)[0]
            body, padded = pos + 8, (size + 1) & ~1
            if tag == b"LIST":
                kind = data[body:body + 4]
                if kind == b"movi":
                    found["movi_size_pos"] = pos + 4
                else:                                     # hdrl / strl: look for indx
                    walk(body + 4, min(body + size, hi), depth + 1)
            elif tag == b"indx" and found["indx"] is None:
                found["indx"] = (body, size)
            elif tag == b"idx1" and depth == 0:
                found["idx1"] = (body, size)
            pos = body + padded

    walk(12, end, 0)

    def std_entries(pos, n, base):
        out, last_off = [], 0
        for i in range(n):
            off, size = struct.unpack_from("<II", data, pos + 8 * i)
            if off == 0:                                  # DataLoader.hx:343-344
                off = last_off
            else:
                last_off = off
            out.append(IndexEntry(base + off - 8, size & 0x7FFFFFFF, (size & 0x80000000) == 0))
        return out

    if found["indx"] is not None:
        body, size = found["indx"]
        longs, _, _, used, ckid = struct.unpack_from("<HBBI4s", data, body)
        if longs == 4:                                    # super index (AVIParser.hx:101-107)
            out: List[IndexEntry] = []
            for i in range(used):
                off, sz, dur = struct.unpack_from("<QII", data, body + 24 + 16 * i)
                if off + 32 > len(data) or data[off:off + 2] != b"ix":
                    return None
                n = struct.unpack_from("<I", data, off + 12)[0]
                base = struct.unpack_from("<Q", data, off + 20)[0]
                part = std_entries(off + 32, n, base)
                out.extend(part[:dur] if dur else part)
            return out
        if longs == 2:                                    # the strl holds a standard index itself (:109-116)
            base = struct.unpack_from("<Q", data, body + 12)[0]
            return std_entries(body + 24, used, base)
    if found["idx1"] is not None and found["movi_size_pos"] is not None:
        body, size = found["idx1"]
        out, first = [], -1
        for i in range(size >> 4):
            cid, flags, off, ln = struct.unpack_from("<IIII", data, body + 16 * i)
            if first < 0:
                first = off
            if (cid & 0xFF0000) == 0x640000:              # '??d?' = video chunk (DataLoaderAVIIndexed.hx:312-315)
                out.append(IndexEntry(off, ln, (flags & 16) > 0))
        # offsets are relative to the 'movi' tag unless they already look absolute (:319-324)
        base = found["movi_size_pos"] + 4 if first < found["movi_size_pos"] else 0
        for e in out:
            e.offset += base
        return out
    return None


def read_avi_indexed(data: bytes) -> Tuple[VideoInfo, List[bytes], Optional[List[bool]]]:
    """Frames fetched through the index (random access, as DataLoaderAVIIndexed does) with the index's key
    flags; an entry of size 0 is an empty frame (DataLoader.hx:389-395).  Without an index: the sequential walk,
    and key flags None — the sequential loader takes them from the decoder, key = (first frame) or
    decoder.IsKeyFrame(bytes) (DataLoaderAVISeq.hx:45), which is what Manager.play does when given None."""
    vi, seq_frames = read_avi(data)
    index = read_index(data)
    if index is None:
        return vi, seq_frames, None
    frames, keys = [], []
    for e in index:
        if e.size == 0:
            frames.append(b"")
        else:
            tag, size = data[e.offset:e.offset + 4], struct.unpack_from("<I", data, e.offset + 4)[0]
            if tag not in (b"00dc", b"00db"):
                raise ValueError(f"index entry at {e.offset} does not point at a video chunk")
            frames.append(data[e.offset + 8:e.offset + 8 + ((size + 1) & ~1)])
        keys.append(e.key)
    return vi, frames, keys


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
