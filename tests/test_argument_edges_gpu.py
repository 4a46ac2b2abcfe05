"""Constructor and argument edge cases through the Python mirror of the C ABI: bad geometry or devices are refused
with a message (jsp_last_error), degenerate-but-legal inputs behave as the reference does, nothing crashes."""
import numpy as np
import pytest

from jsplayer_amd import CodecError, DecoderState, MSVideo1_16bit, MSVideo1_8bit, ScreenPressor

pytestmark = pytest.mark.gpu


def dev(n):
    import torch
    return torch.zeros(n, dtype=torch.int32, device="cuda")


@pytest.mark.parametrize("make,msg", [
    (lambda: MSVideo1_16bit(0, 0), "bad frame size"),
    (lambda: MSVideo1_16bit(-4, 8), "bad frame size"),
    (lambda: ScreenPressor(0, 0, 24), "bad frame size"),
    (lambda: ScreenPressor(100000, 2, 24), "wider than 8192"),
    (lambda: MSVideo1_16bit(16, 16, device=99), "device_id out of range"),
    (lambda: MSVideo1_16bit(1 << 20, 1 << 20), "frame too large"),
])
def test_refused_constructions(make, msg):
    with pytest.raises(CodecError, match=msg):
        make()


def test_degenerate_but_legal_inputs():
    # frames smaller than one 4x4 block: no block is ever painted, the calls still answer as the reference does
    c = MSVideo1_16bit(3, 3)
    c.Preinit(36)
    d = dev(9)
    assert c.DecompressI(b"", d) == DecoderState.zero_state
    assert c.DecompressI(b"\x01\x02\x03\x04", d) == DecoderState.zero_state
    assert not d.cpu().numpy().any()
    with pytest.raises(CodecError, match="width\\*height"):
        c.DecompressI(b"", dev(4))
    # an 8-bit codec with a missing / short palette (the reference reads what is there)
    for pal in (b"", b"\x01\x02\x03"):
        p = MSVideo1_8bit(16, 16, pal)
        p.Preinit(0)
        assert p.DecompressI(b"\x05\x81" * 16, dev(256)) == DecoderState.zero_state
    # ScreenPressor: inter frame before any key frame = "no changes"; empty key frame = error_occured
    s = ScreenPressor(16, 16, 24)
    s.Preinit(-5)
    res = s.DecompressP(b"\x01\x02", dev(256))
    assert res.data_pnt is None and res.significant_changes is False
    assert s.DecompressI(b"", dev(256)) == DecoderState.error_occured
    assert ScreenPressor(16, 16, 7).DecompressI(b"\x12", dev(256)) in (DecoderState.zero_state, DecoderState.error_occured)


@pytest.mark.parametrize("count,size", [(3, (320, 240)), (40, (320, 240)), (33, (322, 242))], ids=["small", "placed", "large-odd-size"])
def test_frame_pool_buffers_are_usable_whichever_way_they_were_placed(count, size):
    """jsp_pool_create: a small pool is one allocation per frame; a pool of 32 frames or more whose size the probe's store shape
    fits is placed by measuring candidate allocations (first: chunks of 16 frames, every fourth of a run of allocations).  Either way every buffer is zeroed, distinct, 16-byte aligned and decodes
    exactly as a torch tensor does."""
    import torch
    from jsplayer_amd import FramePool
    from jsplayer_amd import streamgen as sg
    w, h = size
    pool = FramePool(w, h, count)
    try:
        placed = count >= 32 and w % 4 == 0 and h % 4 == 0
        assert (pool.attempts >= 1 and pool.store_rate > 0) if placed else (pool.attempts == 0 and pool.store_rate == 0)
        assert pool.attempts <= 16
        ptrs = [f.data_ptr() for f in pool.frames]
        assert len(set(ptrs)) == count and all(p % 16 == 0 for p in ptrs)
        spans = sorted(ptrs)
        assert all(b - a >= w * h * 4 for a, b in zip(spans, spans[1:]))          # no two buffers overlap
        assert all(not f.cpu().numpy().any() for f in pool.frames)
        if w % 4 == 0:
            frames, keys, _ = sg.msv1_clip(5, w, h, min(count, 6), p_mix=sg.msv1_p_mix(0.5, 10.0), key_every=3)
            a, b = MSVideo1_16bit(w, h), MSVideo1_16bit(w, h)
            a.Preinit(36)
            b.Preinit(36)
            mine = [dev(w * h) for _ in frames]
            for i, (fr, key) in enumerate(zip(frames, keys)):
                if key:
                    assert a.DecompressI(fr, pool.frames[i]) == b.DecompressI(fr, mine[i])
                else:
                    ra, rb = a.DecompressP(fr, pool.frames[i]), b.DecompressP(fr, mine[i])
                    assert ra.significant_changes == rb.significant_changes
                    assert (ra.data_pnt is pool.frames[i]) == (rb.data_pnt is mine[i])
                if a.PreviousFrame() is not None:
                    assert torch.equal(a.PreviousFrame(), b.PreviousFrame())
            a.StopAndClean()
            b.StopAndClean()
    finally:
        pool.close()


def test_measure_fill_reports_a_store_rate_and_refuses_bad_arguments():
    """jsp_measure_fill (what bench.py prints as roofline.measured_ceiling): a plausible rate for a plain fill, the buffer left
    filled, misaligned / tiny / null arguments refused with a message."""
    import ctypes as C
    import torch
    from jsplayer_amd import _native as N
    lib = N.lib()
    buf = torch.zeros(1 << 24, dtype=torch.int32, device="cuda")          # 64 MiB
    rate = C.c_double(0.0)
    assert lib.jsp_measure_fill(C.c_void_p(buf.data_ptr()), C.c_size_t(buf.numel() * 4), 3, C.byref(rate), None) == 0
    assert 100.0 < rate.value < 20000.0, rate.value                        # GB/s: an MI355X fills at several TB/s; anything sane passes
    assert int(buf[4].item()) != 0 or int(buf[5].item()) != 0              # the fill pattern is v, v+1, v+2, v+3 per 16 bytes
    assert lib.jsp_measure_fill(C.c_void_p(buf.data_ptr() + 4), C.c_size_t(1 << 20), 1, C.byref(rate), None) != 0
    assert "bad argument" in N.last_error()
    assert lib.jsp_measure_fill(C.c_void_p(buf.data_ptr()), C.c_size_t(64), 1, C.byref(rate), None) != 0
    assert lib.jsp_measure_fill(None, C.c_size_t(1 << 20), 1, C.byref(rate), None) != 0


def test_measure_h2d_and_the_bounded_pool_probe(monkeypatch):
    """jsp_measure_h2d (bench.py: e2e.h2d_ceiling_GBs): a plausible rate for pinned host-to-device copies, bad arguments refused.
    jsp_pool_create's placement probe is bounded: JSP_POOL_PROBE_MAX candidates at most, never more held than its limit, and
    jsp_pool_probe_info says what it cost; a small pool is not probed at all."""
    import ctypes as C
    from jsplayer_amd import _native as N
    from jsplayer_amd.codec import FramePool
    lib = N.lib()
    rate = C.c_double(0.0)
    assert lib.jsp_measure_h2d(0, 4 << 20, 2, 4, C.byref(rate)) == 0
    assert 1.0 < rate.value < 200.0, rate.value                 # GB/s: PCIe 5 x16 is 64 GB/s; anything sane passes
    assert lib.jsp_measure_h2d(0, 0, 1, 1, C.byref(rate)) != 0 and lib.jsp_measure_h2d(0, 1 << 20, 0, 1, C.byref(rate)) != 0
    assert lib.jsp_measure_h2d(0, 1 << 20, 17, 1, C.byref(rate)) != 0 and lib.jsp_measure_h2d(0, 1 << 20, 1, 1, None) != 0
    assert lib.jsp_measure_h2d(99, 1 << 20, 1, 1, C.byref(rate)) != 0
    monkeypatch.setenv("JSP_POOL_PROBE_MAX", "2")
    pool = FramePool(1920, 1080, 32)
    try:
        assert 1 <= pool.attempts <= 2 and pool.store_rate > 100.0
        assert pool.probe_ms > 0 and 32 * 1920 * 1080 * 4 <= pool.held_bytes <= 5.1 * 32 * 1920 * 1080 * 4 and pool.held_bytes <= pool.hold_limit   # (at most: the mapped form and a run of four times the pool's chunks)
    finally:
        pool.close()
    monkeypatch.setenv("JSP_POOL_PROBE_HOLD_GB", "0.1")          # less than one candidate: the pool itself is still allowed, nothing beside it
    pool = FramePool(1920, 1080, 32)
    try:
        # (the first form is an address range over physical allocations of 16 frames each, rounded up to 2 MB: the pool and a per cent)
        one = 32 * 1920 * 1080 * 4
        assert pool.attempts == 1 and one == pool.hold_limit and one <= pool.held_bytes <= one * 1.02
    finally:
        pool.close()
    # the form a board liked last time is tried first (here: said through the environment); whatever wins, the pool is whole
    monkeypatch.delenv("JSP_POOL_PROBE_HOLD_GB")
    monkeypatch.setenv("JSP_POOL_PROBE_MAX", "3")
    for form in ("0", "1", "2"):
        monkeypatch.setenv("JSP_POOL_PROBE_FORM", form)
        pool = FramePool(1920, 1080, 40)
        try:
            assert 1 <= pool.attempts <= 3 and len(pool.tried_rates) == pool.attempts and pool.store_rate == max(pool.tried_rates) or pool.store_rate > 0
            ptrs = sorted(f.data_ptr() for f in pool.frames)
            assert len(set(ptrs)) == 40 and all(b - a >= 1920 * 1080 * 4 for a, b in zip(ptrs, ptrs[1:]))
        finally:
            pool.close()
    monkeypatch.delenv("JSP_POOL_PROBE_FORM")
    small = FramePool(320, 240, 9)
    try:
        assert small.attempts == 0 and small.store_rate == 0 and small.probe_ms == 0 and small.held_bytes == 0
    finally:
        small.close()
