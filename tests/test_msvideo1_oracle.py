"""CPU tests: the MSVideo1 oracle against the hand-worked known-answer vectors, against an
independent pure-Python restatement of the CRAM layout, and on the JS edge semantics it claims."""
import json
import os

import numpy as np
import pytest

from jsplayer_amd import streamgen as sg
from oracle_binding import OracleAbort, OracleMSVideo1
import pyref_msv1

HERE = os.path.dirname(os.path.abspath(__file__))
KATS = json.load(open(os.path.join(HERE, "golden", "msv1_kat.json")))


def _why(codec):
    """the native library's description of the last failure (product codecs only)"""
    try:
        from jsplayer_amd import _native
        return _native.last_error() if hasattr(codec, "_h") else ""
    except Exception:
        return ""


def run_kat(make_codec, new_buf, read_buf, kat):
    """Drive one known-answer vector through a codec with the IVideoCodec call protocol."""
    w, h = kat["w"], kat["h"]
    codec = make_codec(kat["bits"], w, h, bytes(kat["palette"]) if "palette" in kat else None)
    codec.Preinit(kat.get("lines", 36))
    bufs = [new_buf(w * h, kat.get("prefill", 0)) for _ in range(3)]
    for i, fr in enumerate(kat["frames"]):
        prev = codec.PreviousFrame()
        dst = next(b for b in bufs if b is not prev)
        src = bytes(fr["src"])
        if fr["key"]:
            assert codec.DecompressI(src, dst) == 0, _why(codec)
            got = codec.PreviousFrame()
        else:
            res = codec.DecompressP(src, dst)
            got, signif = (res.data_pnt, res.significant_changes) if hasattr(res, "data_pnt") else res
            assert signif == fr["signif"], kat["name"]
        if fr["adopted"]:
            assert got is dst, kat["name"]
            assert read_buf(dst).tolist() == [v if v < 2 ** 31 else v - 2 ** 32 for v in fr["expect"]], kat["name"]
        else:
            assert got is prev, kat["name"]


@pytest.mark.parametrize("kat", KATS, ids=[k["name"] for k in KATS])
def test_oracle_known_answers(kat):
    run_kat(lambda bits, w, h, pal: OracleMSVideo1(bits, w, h, pal),
            lambda n, fill: np.full(n, fill, dtype=np.int32), lambda b: b, kat)


@pytest.mark.parametrize("bits", [16, 8])
@pytest.mark.parametrize("size", [(16, 16), (64, 48), (320, 240), (36, 20)])
def test_oracle_matches_python_restatement(bits, size):
    w, h = size
    frames, keys, pal = sg.msv1_clip(100 + bits, w, h, 6, bits=bits, p_mix=sg.msv1_p_mix(0.6, 7.0))
    orc = OracleMSVideo1(bits, w, h, pal)
    orc.Preinit(0)
    pal_ints = pyref_msv1.palette_ints(pal) if pal else None
    bufs = [np.full(w * h, 0x55, dtype=np.int32) for _ in range(2)]
    prev_py = None
    for i, (src, key) in enumerate(zip(frames, keys)):
        dst = bufs[i % 2]
        before = dst.copy()
        if key:
            assert orc.DecompressI(src, dst) == 0
        else:
            data, _ = orc.DecompressP(src, dst)
            assert data is dst
        exp, coded, nskips = pyref_msv1.decode(bits, w, h, src, prev_py, pal_ints, dst=before)
        assert coded
        assert np.array_equal(dst.reshape(h, w).astype(np.int64) & 0xFFFFFFFF, exp & 0xFFFFFFFF)
        prev_py = exp
        assert orc.IsKeyFrame(src) == (nskips == 0)


def test_generator_mix_and_sizes():
    rng = sg.SplitMix64(sg.SEED_BASE + 2)
    f = sg.msv1_frame_16(rng, 1920, 1080, sg.MIX_M1)
    # M1: 25% x 2 + 50% x 6 + 25% x 18 = 8 bytes per block on average
    assert abs(len(f) / 129600 - 8.0) < 0.1
    assert len(sg.msv1_frame_16(rng, 1920, 1080, sg.MIX_ALL_SOLID)) == 259200
    assert len(sg.msv1_frame_16(rng, 1920, 1080, sg.MIX_ALL_EIGHT)) == 2332800
    o = OracleMSVideo1(16, 1920, 1080)
    assert o.IsKeyFrame(f)


def test_truncated_stream_reads_as_js_undefined():
    # 16-bit, one 2-colour code whose second colour is cut off: it reads NaN -> black
    o = OracleMSVideo1(16, 4, 4)
    o.Preinit(0)
    dst = np.full(16, -1, dtype=np.int32)
    assert o.DecompressI(bytes([0x0F, 0x00, 0x1F, 0x00, 0xE0]), dst) == 0
    assert dst.tolist() == [0xF8] * 4 + [0] * 12
    # stream ends before the second block: `a`,`b` undefined -> solid, colour NaN -> 0
    o = OracleMSVideo1(16, 8, 4)
    o.Preinit(0)
    dst = np.full(32, -1, dtype=np.int32)
    o.DecompressI(bytes([0x00, 0xFC] + [0] * 9), dst)  # 11 bytes: not an early-out candidate
    assert dst.reshape(4, 8)[:, :4].tolist() == [[0xF80000] * 4] * 4
    # second code word = bytes 0,0 -> b < 0x80: 2-colour with flags 0xFFFF and colours 0,0
    assert dst.reshape(4, 8)[:, 4:].tolist() == [[0] * 4] * 4
    # odd length: only the low byte of the code word exists -> solid fromRGB15(a)
    o = OracleMSVideo1(16, 4, 4)
    o.Preinit(0)
    dst = np.full(16, -1, dtype=np.int32)
    o.DecompressI(bytes([0x1F]), dst)
    assert dst.tolist() == [0xF8] * 16


def test_skip_before_any_frame_aborts_like_the_reference():
    o = OracleMSVideo1(16, 8, 4)
    o.Preinit(0)
    dst = np.full(32, 7, dtype=np.int32)
    # solid, then skip with prevFrame == null: 12 bytes so the early-out test is not taken
    src = bytes([0x00, 0xFC, 0x01, 0x84] + [0] * 8)
    with pytest.raises(OracleAbort):
        o.DecompressP(src, dst)
    assert o.PreviousFrame() is None
    assert dst.reshape(4, 8)[:, :4].tolist() == [[0xF80000] * 4] * 4
    assert dst.reshape(4, 8)[:, 4:].tolist() == [[7] * 4] * 4


def test_negative_skip_copies_the_rest_of_the_frame():
    w, h = 16, 8
    frames, _, _ = sg.msv1_clip(7, w, h, 1)
    o = OracleMSVideo1(16, w, h)
    o.Preinit(0)
    a, b = np.zeros(w * h, np.int32), np.full(w * h, 5, np.int32)
    o.DecompressI(frames[0], a)
    # one solid block, then skip code with count 0 (skip = -1): everything else copied
    data, signif = o.DecompressP(bytes([0x1F, 0x80, 0x00, 0x84] + [0] * 8), b)
    assert data is b
    exp = a.copy().reshape(h, w)
    exp[:4, :4] = 0xF8
    assert np.array_equal(b.reshape(h, w), exp)


def test_8bit_significance_quirk():
    w, h = 16, 48
    frames, keys, pal = sg.msv1_clip(9, w, h, 3, bits=8, p_mix=sg.msv1_p_mix(0.3, 3.0))
    o = OracleMSVideo1(8, w, h, pal)
    o.Preinit(4)
    bufs = [np.zeros(w * h, np.int32) for _ in range(2)]
    o.DecompressI(frames[0], bufs[0])
    # with a previous frame the 8-bit class never reports significant changes (NaN loop bound)
    data, signif = o.DecompressP(frames[1], bufs[1])
    assert data is bufs[1] and signif is False
    # without one, stage 1 alone decides
    o2 = OracleMSVideo1(8, w, h, pal)
    o2.Preinit(4)
    data, signif = o2.DecompressP(frames[0], bufs[0])
    assert signif is True


def test_product_host_parser_agrees_with_oracle_on_cpu():
    """msv1_host.cpp (the product's sequential parser) through the test shim: IsKeyFrame, the
    no-pixel facts (early-out, changes, abort) and the block counts against the oracle, on valid,
    truncated and random streams.  (Pixels are compared on the GPU.)"""
    import ctypes as C
    import hoststage_binding as hs
    L = hs.lib()
    L.hs_msv1_parse.argtypes = [C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.hs_msv1_is_key.argtypes = [C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_size_t]
    rng = np.random.default_rng(21)
    checked = 0
    for bits in (16, 8):
        for (w, h) in [(16, 8), (37, 23), (64, 48)]:
            frames, keys, pal = sg.msv1_clip(8200 + bits, w, h, 4, bits=bits, p_mix=sg.msv1_p_mix(0.5, 5.0))
            nblocks, nby = (w >> 2) * (h >> 2), h >> 2
            orc = OracleMSVideo1(bits, w, h, pal)
            orc.Preinit(4)
            bufs = [np.zeros(w * h, np.int32) for _ in range(3)]
            bc = np.zeros(max(nby, 1), np.uint8)
            for k in range(150):
                b = bytearray(frames[k % 4])
                if k % 3 == 1 and b:
                    b = b[: int(rng.integers(0, len(b)))]
                if k % 3 == 2:
                    b = bytearray(rng.integers(0, 256, size=int(rng.integers(0, 80)), dtype=np.uint8).tobytes())
                src = bytes(b)
                assert bool(L.hs_msv1_is_key(bits, w, h, src, len(src))) == orc.IsKeyFrame(src)
                have_prev = orc.PreviousFrame() is not None
                desc = np.zeros(max(nblocks, 1), np.uint32)
                out = np.zeros(8, np.uint64)
                L.hs_msv1_parse(bits, w, h, src, len(src), int(have_prev), 4, desc.ctypes.data, bc.ctypes.data, out.ctypes.data)
                early, changes, s1, aborted = (bool(v) for v in out[:4])
                dst = next(x for x in bufs if x is not orc.PreviousFrame())
                try:
                    data, sig = orc.DecompressP(src, dst)
                    assert not aborted
                    assert (data is dst) == changes
                    if not early:
                        assert int(out[4] + out[5] + out[6]) == nblocks
                    if not have_prev:
                        assert sig == s1      # without a previous frame stage 1 decides alone
                except OracleAbort:
                    assert aborted
                checked += 1
    assert checked == 900
