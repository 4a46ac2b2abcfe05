"""The lane-private part of the on-GPU MSVideo1 parse (jsplayer_amd/csrc/msv1_lanes.h) against a sequential walk.

The header is compiled for the host by tests/lanes (its GPU instructions emulated); here every lane result — slot
masks, the 9-entry table "entry slot -> (exit slot, blocks)", the visited-slot sets — is compared with a plain
Python walk over the same words following the code-length rules of MSVideo1.hx:128-181 (16-bit) / :311-364 (8-bit)
as SURVEY.md Appendix A states them."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REST = 0xFFFFF


@pytest.fixture(scope="module")
def L():
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "lanes")])
    lib = C.CDLL(os.path.join(HERE, "lanes", "liblanes.so"))
    lib.lanes_one.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]
    lib.lanes_perm.restype = C.c_uint32
    lib.lanes_perm.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
    return lib


def classify(bits, words, s, nvalid):
    """(length in slots, blocks covered, kind) of the code that would start at slot s; words[s+1] is readable."""
    if s >= nvalid:
        return 1, 0, "invalid"
    w = int(words[s])
    b = w >> 8
    if (b & 0xFC) == 0x84:
        n = w & 0x3FF
        return 1, (n if n else REST), "skip"
    if bits == 16:
        if b < 0x80:
            return (9 if (int(words[s + 1]) & 0x8000) else 3), 1, "coded"
        return 1, 1, "coded"
    if w == 0:
        return 1, 0, "end"
    if b < 0x80:
        return 2, 1, "coded"
    if b >= 0x90:
        return 5, 1, "coded"
    return 1, 1, "coded"


def walk(bits, ls, words, nvalid, entry):
    pos, blocks, seen = entry, 0, 0
    while pos < ls:
        ln, cnt, _ = classify(bits, words, pos, nvalid)
        seen |= 1 << pos
        blocks += cnt
        pos += ln
    return pos - ls, blocks, seen


def random_words(rng, bits, ls, style):
    n = ls + 1
    if style == "uniform":
        w = rng.integers(0, 1 << 16, n)
    elif style == "solid":          # high byte >= 0x80: one-slot codes, some of them skip codes
        w = rng.integers(0x8000, 1 << 16, n)
    elif style == "pattern":        # high byte < 0x80
        w = rng.integers(0, 0x8000, n)
    elif style == "skips":
        w = rng.integers(0x8400, 0x8800, n)
        w[rng.random(n) < 0.2] = 0x8400   # count 0 = the rest of the frame
    elif style == "zeros":
        w = np.zeros(n, dtype=np.int64)
        k = rng.random(n) < 0.3
        w[k] = rng.integers(0, 1 << 16, int(k.sum()))
    else:                            # a plausible mix of the four
        w = rng.integers(0, 1 << 16, n)
        pick = rng.random(n)
        w[pick < 0.3] &= 0x7FFF
        w[pick > 0.9] = rng.integers(0x8400, 0x8800, int((pick > 0.9).sum()))
        if bits == 8:
            w[(pick > 0.5) & (pick < 0.6)] |= 0x9000
    return w.astype(np.uint32)


def run_lane(L, bits, ls, words, nvalid, zw_all):
    vis = np.array(words, dtype=np.uint32).copy()
    vis[min(nvalid, ls + 1):] = 0            # the kernel zeroes the bytes past the frame's data
    packed = (vis[0::2][: ls // 2 + 1].astype(np.uint32) | (np.append(vis[1::2], 0)[: ls // 2 + 1].astype(np.uint32) << 16)).astype(np.uint32)
    out = np.zeros(24, dtype=np.uint32)
    L.lanes_one(bits, ls, packed.ctypes.data, nvalid, zw_all, out.ctypes.data)
    return vis, out


def test_perm_sign_selectors(L):
    # V_PERM_B32 selectors 8..11: bit 7 of bytes 1, 3, 5, 7 of {s0, s1}, replicated
    assert L.lanes_perm(0x0080FF00, 0x80000000, 0x0B0A0908) == 0x00FFFF00    # bytes 1, 3 of s1 = 00, 80; of s0 = FF, 00
    assert L.lanes_perm(0, 0x00008000, 0x0B0A0908) == 0x000000FF
    assert L.lanes_perm(0x12345678, 0x9ABCDEF0, 0x07060100) == 0x1234DEF0


@pytest.mark.parametrize("bits", [16, 8])
@pytest.mark.parametrize("ls", [32, 16])
def test_lane_tables_and_visited_sets_match_the_sequential_walk(L, bits, ls):
    rng = np.random.default_rng(1234 + bits + ls)
    styles = ["uniform", "solid", "pattern", "skips", "zeros", "mix"]
    n_lanes = 0
    for it in range(1500):
        style = styles[it % len(styles)]
        words = random_words(rng, bits, ls, style)
        r = rng.random()
        nvalid = ls + 1 if r < 0.6 else int(rng.integers(0, ls + 2))      # lanes of a frame's last tile: data ends inside
        vis, out = run_lane(L, bits, ls, words, nvalid, zw_all=it % 2)
        nv = min(nvalid, ls)
        allm = (1 << ls) - 1
        # masks
        M = Lm = Z = K = EM = 0
        for s in range(ls):
            ln, cnt, kind = classify(bits, vis, s, nv)
            if ln in (3, 2):
                M |= 1 << s
            elif ln in (9, 5):
                Lm |= 1 << s
            if kind in ("skip", "invalid", "end"):
                Z |= 1 << s
            if kind == "skip":
                K |= 1 << s
            if kind == "end":
                EM |= 1 << s
        assert (int(out[0]), int(out[1]), int(out[2]), int(out[3]), int(out[4])) == (M, Lm, Z, K, EM), (style, nvalid)
        assert int(out[5]) == ((1 << nv) - 1 if nv < ls else allm)
        # tables and visited sets
        for e in range(9):
            if e >= ls:
                continue
            ex, blocks, seen = walk(bits, ls, vis, nv, e)
            got = int(out[6 + e])
            if blocks >= (1 << 28):
                assert got >> 4 >= (1 << 28) - 1
            else:
                assert got == (ex | (blocks << 4)), (style, e, nvalid, hex(got), ex, blocks)
            assert int(out[15 + e]) == seen & allm, (style, e)
        n_lanes += 1
    assert n_lanes == 1500


def test_lane_table_saturates_instead_of_wrapping(L):
    # 32 skip codes of count 0 ("the rest of the frame", 0xFFFFF blocks each): 32 * 0xFFFFF < 2^28, so add lanes' worth by hand:
    # the saturating add must never wrap the packed value
    words = np.full(33, 0x8400, dtype=np.uint32)
    vis, out = run_lane(L, 16, 32, words, 33, zw_all=1)
    assert int(out[6]) == (0 | ((32 * REST) << 4))
