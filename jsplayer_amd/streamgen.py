"""Synthetic block-stream generators (the reference ships no encoder and no sample media,
SURVEY.md §4).  Inputs follow SURVEY.md §8(d): splitmix64, seed 0x6A73706C00000000 + config.

MSVideo1 bit layout as the reference decodes it: SURVEY.md Appendix A / MSVideo1.hx:128-181,311-364.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

SEED_BASE = 0x6A73706C00000000
_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


class SplitMix64:
    """Counter-based splitmix64: the i-th output depends only on (seed, i), so blocks of outputs
    can be produced with numpy."""

    def __init__(self, seed: int):
        self.state = np.uint64(seed & 0xFFFFFFFFFFFFFFFF)

    def u64(self, n: int) -> np.ndarray:
        with np.errstate(over="ignore"):
            idx = np.arange(1, n + 1, dtype=np.uint64)
            z = self.state + idx * _GAMMA
            self.state = self.state + np.uint64(n) * _GAMMA
            z = (z ^ (z >> np.uint64(30))) * _M1
            z = (z ^ (z >> np.uint64(27))) * _M2
            return z ^ (z >> np.uint64(31))

    def uniform(self, n: int) -> np.ndarray:
        return (self.u64(n) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))

    def below(self, n: int, bound: int) -> np.ndarray:
        return (self.u64(n) % np.uint64(bound)).astype(np.int64)


# code kinds
SKIP, SOLID, TWO, EIGHT = 0, 1, 2, 3


@dataclass
class Msv1Mix:
    """Probabilities of the coded-block kinds and of starting a skip run."""
    solid: float = 0.25
    two: float = 0.50
    eight: float = 0.25
    skip_start: float = 0.0     # probability that a code is a skip code
    skip_mean: float = 40.0     # mean length of a skip run (geometric, clipped to 1..1023)


MIX_M1 = Msv1Mix()                                   # 25 % solid / 50 % 2-colour / 25 % 8-colour
MIX_ALL_SOLID = Msv1Mix(1.0, 0.0, 0.0)
MIX_ALL_EIGHT = Msv1Mix(0.0, 0.0, 1.0)


def msv1_p_mix(skipped_fraction: float = 0.70, skip_mean: float = 40.0) -> Msv1Mix:
    """Mix for inter frames: about `skipped_fraction` of the blocks covered by skip runs."""
    # a skip code covers skip_mean blocks, a coded one covers 1: solve for the code probability
    f, m = skipped_fraction, skip_mean
    p = f / (f + m * (1.0 - f)) if f < 1.0 else 1.0
    return Msv1Mix(skip_start=p, skip_mean=m)


def _plan_codes(rng: SplitMix64, nblocks: int, mix: Msv1Mix) -> Tuple[np.ndarray, np.ndarray]:
    """Draw a sequence of codes covering exactly `nblocks` blocks -> (kind[], blocks_covered[])."""
    u = rng.uniform(nblocks)
    coded = np.array([mix.solid, mix.two, mix.eight], dtype=np.float64)
    coded = coded / coded.sum()
    edges = np.cumsum(coded) * (1.0 - mix.skip_start) + mix.skip_start
    kind = np.full(nblocks, EIGHT, dtype=np.int64)
    kind[u < edges[1]] = TWO
    kind[u < edges[0]] = SOLID
    kind[u < mix.skip_start] = SKIP
    cover = np.ones(nblocks, dtype=np.int64)
    nskip = int((kind == SKIP).sum())
    if nskip:
        g = rng.uniform(nskip)
        run = np.floor(np.log1p(-g) / np.log1p(-1.0 / max(mix.skip_mean, 1.0001))).astype(np.int64) + 1
        cover[kind == SKIP] = np.clip(run, 1, 1023)
    end = np.cumsum(cover)
    ncodes = int(np.searchsorted(end, nblocks, side="left")) + 1
    kind, cover, end = kind[:ncodes], cover[:ncodes], end[:ncodes]
    over = int(end[-1]) - nblocks
    if over > 0:
        cover[-1] -= over  # only a skip run can overshoot
    return kind, cover


def msv1_frame_16(rng: SplitMix64, width: int, height: int, mix: Msv1Mix = MIX_M1) -> bytes:
    """One 16-bit (RGB555) MSVideo1 frame as raw chunk bytes."""
    nblocks = (width >> 2) * (height >> 2)
    if nblocks == 0:
        return b""
    kind, cover = _plan_codes(rng, nblocks, mix)
    n = kind.size
    size = np.array([2, 2, 6, 18], dtype=np.int64)[kind]
    off = np.concatenate(([0], np.cumsum(size)[:-1]))
    total = int(size.sum())
    out = np.zeros(total // 2, dtype=np.uint16)
    w = (off // 2)
    r = rng.u64(n * 9).reshape(n, 9)
    word = (r & np.uint64(0xFFFF)).astype(np.uint16)

    sk = kind == SKIP
    out[w[sk]] = (0x8400 + cover[sk]).astype(np.uint16)          # b = 0x84 + (n >> 8), a = n & 0xFF

    so = kind == SOLID
    c = word[so, 0] | np.uint16(0x8000)
    clash = (c & np.uint16(0xFC00)) == np.uint16(0x8400)            # would read as a skip code
    c = np.where(clash, c ^ np.uint16(0x1000), c)
    out[w[so]] = c

    for k, ncol in ((TWO, 2), (EIGHT, 8)):
        m = kind == k
        if not m.any():
            continue
        base = w[m]
        out[base] = word[m, 0] & np.uint16(0x7FFF)                  # flags, high byte < 0x80
        first = word[m, 1]
        first = (first | np.uint16(0x8000)) if k == EIGHT else (first & np.uint16(0x7FFF))
        out[base + 1] = first
        for j in range(1, ncol):
            out[base + 1 + j] = word[m, 1 + j]
    return out.astype("<u2").tobytes()


def msv1_frame_8(rng: SplitMix64, width: int, height: int, mix: Msv1Mix = MIX_M1) -> bytes:
    """One 8-bit (palettised) MSVideo1 frame as raw chunk bytes."""
    nblocks = (width >> 2) * (height >> 2)
    if nblocks == 0:
        return b""
    kind, cover = _plan_codes(rng, nblocks, mix)
    n = kind.size
    size = np.array([2, 2, 4, 10], dtype=np.int64)[kind]
    off = np.concatenate(([0], np.cumsum(size)[:-1]))
    out = np.zeros(int(size.sum()), dtype=np.uint8)
    r = rng.u64(n * 2).reshape(n, 2)
    lo = (r[:, 0] & np.uint64(0xFF)).astype(np.uint8)
    hi = ((r[:, 0] >> np.uint64(8)) & np.uint64(0xFF)).astype(np.uint8)
    idx = np.frombuffer(r[:, 1].tobytes(), dtype=np.uint8).reshape(n, 8)

    sk = kind == SKIP
    out[off[sk]] = (cover[sk] & 0xFF).astype(np.uint8)
    out[off[sk] + 1] = (0x84 + (cover[sk] >> 8)).astype(np.uint8)

    so = kind == SOLID
    out[off[so]] = lo[so]
    b = (hi[so] & np.uint8(0x0F)) | np.uint8(0x80)                 # 0x80..0x8F
    b = np.where((b & np.uint8(0xFC)) == np.uint8(0x84), b ^ np.uint8(0x08), b)
    out[off[so] + 1] = b

    tw = kind == TWO
    a2, b2 = lo[tw], hi[tw] & np.uint8(0x7F)
    a2 = np.where((a2 == 0) & (b2 == 0), np.uint8(1), a2)          # 0,0 would be the end marker
    out[off[tw]] = a2
    out[off[tw] + 1] = b2
    out[off[tw] + 2] = idx[tw, 0]
    out[off[tw] + 3] = idx[tw, 1]

    ei = kind == EIGHT
    b8 = hi[ei] | np.uint8(0x90)                                    # >= 0x90
    out[off[ei]] = lo[ei]
    out[off[ei] + 1] = b8
    for j in range(8):
        out[off[ei] + 2 + j] = idx[ei, j]
    return out.tobytes()


def random_palette(rng: SplitMix64, entries: int = 256) -> bytes:
    """`entries` RGBQUADs (B,G,R,0), as found after the BITMAPINFOHEADER in strf."""
    v = rng.u64(entries)
    q = np.zeros((entries, 4), dtype=np.uint8)
    q[:, 0] = (v & np.uint64(0xFF)).astype(np.uint8)
    q[:, 1] = ((v >> np.uint64(8)) & np.uint64(0xFF)).astype(np.uint8)
    q[:, 2] = ((v >> np.uint64(16)) & np.uint64(0xFF)).astype(np.uint8)
    return q.tobytes()


def msv1_clip(config_index: int, width: int, height: int, nframes: int, bits: int = 16,
              key_mix: Msv1Mix = MIX_M1, p_mix: Optional[Msv1Mix] = None,
              key_every: int = 0) -> Tuple[List[bytes], List[bool], Optional[bytes]]:
    """A clip: frame 0 fully coded; later frames are key frames (fully coded) when `p_mix` is None
    or every `key_every` frames, inter frames (with skip runs) otherwise.
    Returns (frames, is_key, palette)."""
    rng = SplitMix64(SEED_BASE + config_index)
    palette = random_palette(rng) if bits == 8 else None
    gen = msv1_frame_16 if bits == 16 else msv1_frame_8
    frames, keys = [], []
    for i in range(nframes):
        key = i == 0 or p_mix is None or (key_every > 0 and i % key_every == 0)
        frames.append(gen(rng, width, height, key_mix if key else p_mix))
        keys.append(key)
    return frames, keys, palette


# ================================================================= ScreenPressor =================
# Encoder = jsplayer_amd/libjspgen.so (gen/sp_encoder.cpp).  The synthetic "desktop" content follows
# flat background, filled rectangles, gradient rectangles, noise patches;
# inter frames move / repaint a few 16x16 blocks.

import ctypes as _C
import os as _os

_GEN_PATH = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "libjspgen.so")
_gen = None


def _genlib():
    global _gen
    if _gen is None:
        if not _os.path.exists(_GEN_PATH):
            raise ImportError(f"{_GEN_PATH} not found: run `make -C jsplayer_amd/gen`")
        L = _C.CDLL(_GEN_PATH)
        L.jspgen_sp_create.restype = _C.c_void_p
        L.jspgen_sp_create.argtypes = [_C.c_int, _C.c_int, _C.c_int, _C.c_int]
        L.jspgen_sp_destroy.argtypes = [_C.c_void_p]
        L.jspgen_sp_error.restype = _C.c_char_p
        L.jspgen_sp_error.argtypes = [_C.c_void_p]
        for name in ("jspgen_sp_encode_i", "jspgen_sp_encode_p", "jspgen_sp_encode_flat"):
            getattr(L, name).restype = _C.c_long
        L.jspgen_sp_encode_i.argtypes = [_C.c_void_p, _C.c_void_p, _C.c_void_p, _C.c_size_t]
        L.jspgen_sp_encode_flat.argtypes = [_C.c_void_p, _C.c_uint32, _C.c_void_p, _C.c_size_t]
        L.jspgen_sp_encode_p.argtypes = [_C.c_void_p, _C.c_void_p, _C.c_void_p, _C.c_void_p, _C.c_size_t]
        L.jspgen_sp_current.argtypes = [_C.c_void_p, _C.c_void_p]
        L.jspgen_sp_set_stale.argtypes = [_C.c_void_p, _C.c_void_p]
        _gen = L
    return _gen


class SpEncoder:
    """Lossless ScreenPressor encoder, stream version 2 (range coder), 3 or 4 (rANS)."""

    def __init__(self, width: int, height: int, bpp: int = 24, version: int = 4):
        self.L = _genlib()
        self.X, self.Y = width, height
        self.h = self.L.jspgen_sp_create(width, height, bpp, version)
        if not self.h:
            raise ValueError("bad encoder parameters")
        self._buf = np.empty(width * height * 6 + 4096, dtype=np.uint8)

    def _ret(self, n):
        if n < 0:
            raise RuntimeError(self.L.jspgen_sp_error(self.h).decode() or f"output buffer too small ({-n})")
        return self._buf[:n].tobytes()

    def encode_i(self, frame: np.ndarray) -> bytes:
        f = np.ascontiguousarray(frame, dtype=np.uint32).reshape(-1)
        assert f.size == self.X * self.Y
        return self._ret(self.L.jspgen_sp_encode_i(self.h, f.ctypes.data, self._buf.ctypes.data, self._buf.size))

    def encode_flat(self, colour: int) -> bytes:
        return self._ret(self.L.jspgen_sp_encode_flat(self.h, colour, self._buf.ctypes.data, self._buf.size))

    def encode_p(self, frame: np.ndarray, hints: Optional[np.ndarray] = None) -> bytes:
        f = np.ascontiguousarray(frame, dtype=np.uint32).reshape(-1)
        assert f.size == self.X * self.Y
        hp = None
        if hints is not None:
            hints = np.ascontiguousarray(hints, dtype=np.int16)
            hp = hints.ctypes.data
        return self._ret(self.L.jspgen_sp_encode_p(self.h, f.ctypes.data, hp, self._buf.ctypes.data, self._buf.size))

    def set_stale(self, picture: Optional[np.ndarray]) -> None:
        """Tests only: tell the encoder what the decoder's destination buffer will hold before the next inter frame, so
        that it may code a column-0 pixel as "the pixel to the left" (ScreenPressor.hx:436-444 reads the last pixel of the
        row above — stale content of the destination — there).  None: back to never emitting that."""
        if picture is None:
            self.L.jspgen_sp_set_stale(self.h, None)
            return
        f = np.ascontiguousarray(picture, dtype=np.uint32).reshape(-1)
        assert f.size == self.X * self.Y
        self.L.jspgen_sp_set_stale(self.h, f.ctypes.data)

    def current(self) -> np.ndarray:
        """The frame a decoder holds after the last encoded frame."""
        out = np.empty(self.X * self.Y, dtype=np.uint32)
        self.L.jspgen_sp_current(self.h, out.ctypes.data)
        return out

    def close(self):
        if self.h:
            self.L.jspgen_sp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


NO_HINT = -32768


def _colours(rng: SplitMix64, n: int, bpp: int) -> np.ndarray:
    v = rng.u64(n)
    if bpp == 16:  # 5-bit components, one per byte (Manager.hx:325-390 shifts them left by 3 for display)
        return ((v & np.uint64(0x1F)) | ((v >> np.uint64(8)) & np.uint64(0x1F)) << np.uint64(8) |
                ((v >> np.uint64(16)) & np.uint64(0x1F)) << np.uint64(16)).astype(np.uint32)
    return (v & np.uint64(0xFFFFFF)).astype(np.uint32)


def desktop_frame(rng: SplitMix64, width: int, height: int, bpp: int = 24, rects: int = 200, gradients: int = 40,
                  noise: float = 0.05) -> np.ndarray:
    """Synthetic screen content (uint32 0x00RRGGBB, shape (height, width))."""
    img = np.full((height, width), int(_colours(rng, 1, bpp)[0]), dtype=np.uint32)
    scale = max(1, (width * height) // (1920 * 1080 // 4))
    nr, ng = max(2, rects * min(scale, 4) // 4), max(1, gradients * min(scale, 4) // 4)
    maxv = 31 if bpp == 16 else 255
    pos = rng.u64(4 * (nr + ng))
    cols = _colours(rng, nr + ng, bpp)
    k = 0
    for i in range(nr):
        x0, y0 = int(pos[k] % np.uint64(width)), int(pos[k + 1] % np.uint64(height))
        w, h = 1 + int(pos[k + 2] % np.uint64(max(2, width // 4))), 1 + int(pos[k + 3] % np.uint64(max(2, height // 4)))
        k += 4
        img[y0:y0 + h, x0:x0 + w] = cols[i]
    for i in range(ng):
        x0, y0 = int(pos[k] % np.uint64(width)), int(pos[k + 1] % np.uint64(height))
        w, h = 2 + int(pos[k + 2] % np.uint64(max(3, width // 6))), 2 + int(pos[k + 3] % np.uint64(max(3, height // 6)))
        k += 4
        yy, xx = np.mgrid[0:min(h, height - y0), 0:min(w, width - x0)]
        base = int(cols[nr + i])
        r = ((base & 0xFF) + xx) % (maxv + 1)
        g = (((base >> 8) & 0xFF) + yy) % (maxv + 1)
        b = (((base >> 16) & 0xFF) + xx + yy) % (maxv + 1)
        img[y0:y0 + yy.shape[0], x0:x0 + xx.shape[1]] = (r | (g << 8) | (b << 16)).astype(np.uint32)
    # noise patches covering about `noise` of the area
    npatch = max(1, int(noise * width * height / (32 * 32)))
    pp = rng.u64(2 * npatch)
    for i in range(npatch):
        x0, y0 = int(pp[2 * i] % np.uint64(width)), int(pp[2 * i + 1] % np.uint64(height))
        h, w = min(32, height - y0), min(32, width - x0)
        img[y0:y0 + h, x0:x0 + w] = _colours(rng, h * w, bpp).reshape(h, w)
    return img


def desktop_next(rng: SplitMix64, img: np.ndarray, bpp: int = 24, p_motion: float = 0.04, p_data: float = 0.02,
                 p_sub: float = 0.02, max_mv: int = 32) -> Tuple[np.ndarray, np.ndarray]:
    """Next frame of a clip: most 16x16 blocks unchanged, some moved (motion vector hint returned),
    some repainted, some changed inside a sub-rectangle.  Returns (frame, hints[nblocks, 2])."""
    height, width = img.shape
    nbx, nby = (width + 15) // 16, (height + 15) // 16
    out = img.copy()
    hints = np.full((nbx * nby, 2), NO_HINT, dtype=np.int16)
    u = rng.uniform(nbx * nby)
    r = rng.u64(nbx * nby * 4).reshape(-1, 4)
    last_mv = (0, 0)
    for bi in np.nonzero(u < p_motion + p_data + p_sub)[0]:
        by, bx = divmod(int(bi), nbx)
        x0, y0 = bx * 16, by * 16
        x1, y1 = min(x0 + 16, width), min(y0 + 16, height)
        if u[bi] < p_motion:
            if int(r[bi, 2] % np.uint64(10)) < 3:
                mx, my = last_mv
            else:
                mx = int(r[bi, 0] % np.uint64(2 * max_mv + 1)) - max_mv
                my = int(r[bi, 1] % np.uint64(2 * max_mv + 1)) - max_mv
            if x0 + mx < 0 or y0 + my < 0 or x1 + mx > width or y1 + my > height or (mx == 0 and my == 0):
                continue
            out[y0:y1, x0:x1] = img[y0 + my:y1 + my, x0 + mx:x1 + mx]
            hints[bi] = (mx, my)
            last_mv = (mx, my)
        elif u[bi] < p_motion + p_data:
            kind = int(r[bi, 0] % np.uint64(3))
            if kind == 0:
                out[y0:y1, x0:x1] = _colours(rng, (y1 - y0) * (x1 - x0), bpp).reshape(y1 - y0, x1 - x0)
            elif kind == 1:
                out[y0:y1, x0:x1] = int(_colours(rng, 1, bpp)[0])
            else:
                yy, xx = np.mgrid[0:y1 - y0, 0:x1 - x0]
                m = 31 if bpp == 16 else 255
                out[y0:y1, x0:x1] = (((xx * 3) % (m + 1)) | (((yy * 5) % (m + 1)) << 8) | (((xx + yy) % (m + 1)) << 16)).astype(np.uint32)
        else:
            if x1 - x0 < 4 or y1 - y0 < 4:
                continue
            sx0 = x0 + 1 + int(r[bi, 0] % np.uint64(x1 - x0 - 3))
            sy0 = y0 + 1 + int(r[bi, 1] % np.uint64(y1 - y0 - 3))
            sx1 = min(x1 - 1, sx0 + 1 + int(r[bi, 2] % np.uint64(8)))
            sy1 = min(y1 - 1, sy0 + 1 + int(r[bi, 3] % np.uint64(8)))
            out[sy0:sy1, sx0:sx1] = _colours(rng, (sy1 - sy0) * (sx1 - sx0), bpp).reshape(sy1 - sy0, sx1 - sx0)
    return out, hints


def sp_clip(config_index: int, width: int, height: int, nframes: int, bpp: int = 24, version: int = 4,
            key_every: int = 0, flat_at: Sequence[int] = (), unchanged_at: Sequence[int] = (),
            p_mix_at: Optional[dict] = None, **frame_kw) -> Tuple[List[bytes], List[bool], List[np.ndarray]]:
    """Encode a synthetic clip.  Returns (chunks, is_key, expected frames as uint32 arrays).
    `p_mix_at` = {frame index: dict(unchanged=..., motion=...)} overrides the block mix of single inter
    frames (the rest of the blocks splits evenly into repainted and sub-rectangle blocks)."""
    rng = SplitMix64(SEED_BASE + config_index)
    enc = SpEncoder(width, height, bpp, version)
    chunks, keys, frames = [], [], []
    img = desktop_frame(rng, width, height, bpp, **frame_kw)
    for i in range(nframes):
        if i in flat_at and i > 0:
            chunks.append(enc.encode_flat(int(_colours(rng, 1, bpp)[0])))
            keys.append(True)
            img = enc.current().reshape(height, width).copy()
        elif i == 0 or (key_every and i % key_every == 0):
            if i > 0:
                img = desktop_frame(rng, width, height, bpp, **frame_kw)
            chunks.append(enc.encode_i(img))
            keys.append(True)
        elif i in unchanged_at:
            chunks.append(enc.encode_p(img))
            keys.append(False)
        else:
            mix = (p_mix_at or {}).get(i)
            if mix:
                rest = max(0.0, 1.0 - mix["unchanged"] - mix["motion"])
                img, hints = desktop_next(rng, img, bpp, p_motion=mix["motion"], p_data=rest / 2, p_sub=rest / 2)
            else:
                img, hints = desktop_next(rng, img, bpp)
            chunks.append(enc.encode_p(img, hints))
            keys.append(False)
        frames.append(img.reshape(-1).copy())
    enc.close()
    return chunks, keys, frames
