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
