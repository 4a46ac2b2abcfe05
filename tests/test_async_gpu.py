"""The asynchronous per-frame calls (jsp_decompress_i_async / _p_async ... jsp_wait; run with -m gpu): with several
frames in flight every frame must come back exactly as the synchronous call — and the CPU oracle — deliver it: same
DecoderState / PFrameResult (buffer identity, significant_changes), same pixels, including the frames the GPU cannot
settle alone and hands back to the synchronous path (truncated streams, 8-bit end markers, skip codes with nothing
to copy from)."""
import numpy as np
import pytest

from jsplayer_amd import CodecError, DecoderState, HostBuffer, MSVideo1_16bit, MSVideo1_8bit, ScreenPressor
from jsplayer_amd import streamgen as sg
from oracle_binding import OracleAbort, OracleMSVideo1, OracleScreenPressor

pytestmark = pytest.mark.gpu


def drive(gpu, orc, w, h, chunks, keys, depth=4, pinned=False, lines=36, before_close=None, prefetch=0, drop_ranges_at=None):
    import torch
    gpu.Preinit(lines)
    orc.Preinit(lines)
    gpu.set_option("async_depth", str(depth))
    nbuf = 2 * depth + 3   # in flight: a destination each, plus the picture each frame is checked against
    gbufs = [torch.full((w * h,), 0x00A5A5A5, dtype=torch.int32, device="cuda") for _ in range(nbuf)]
    obufs = [np.full(w * h, 0x00A5A5A5, dtype=np.int32) for _ in range(nbuf)]
    torch.cuda.synchronize()
    arena = None
    spans = []
    if pinned:   # the compressed frames live in pinned memory: uploaded from where they are
        gap = 8 if prefetch else 0                            # (as in a file: a chunk header in front of every frame, frames at even offsets only)
        if pinned == "pageable-arena":                        # one ordinary allocation holding the file (jsp_prefetch takes ranges of it up all the same)
            class _Plain:
                def __init__(self, n):
                    self.array = np.zeros(n, dtype=np.uint8)

                def close(self):
                    pass
            arena = _Plain(sum(len(c) + gap + 1 for c in chunks) + 64)
        else:
            arena = HostBuffer(sum(len(c) + gap + 1 for c in chunks) + 64)
        pos, srcs = 2 if prefetch else 0, []
        for c in chunks:
            pos += gap
            arena.array[pos:pos + len(c)] = np.frombuffer(c, dtype=np.uint8)
            srcs.append(arena.array[pos:pos + len(c)])
            spans.append((pos, pos + len(c)))
            pos += len(c) + (len(c) & 1 if prefetch else 0)
    else:
        srcs = list(chunks)
    # prefetch = R: the arena goes up in ranges of R frames (jsp_prefetch), the range after the current one ahead of it; every third
    # range stops 5 bytes short of its last frame, which then has to find its own way up
    fetched = set()

    def fetch_range(r):
        if r in fetched or r * prefetch >= len(chunks):
            return
        fetched.add(r)
        lo = spans[r * prefetch][0] - 8
        hi = spans[min((r + 1) * prefetch, len(chunks)) - 1][1] - (5 if r % 3 == 2 else 0)
        gpu.prefetch(arena.array[lo:hi])
    inflight = []   # (ticket, frame index, buffer index, oracle result, oracle picture)

    def collect():
        ticket, i, k, want, picture = inflight.pop(0)
        if want == "raise":
            with pytest.raises(CodecError):
                gpu.wait(ticket)
            return
        got = gpu.wait(ticket)
        if keys[i]:
            assert (got == DecoderState.zero_state) == (want == 0), f"frame {i}: {got} vs oracle {want}"
        else:
            odata_idx, osig = want
            assert got.significant_changes == osig, f"frame {i}"
            gi = next((j for j in range(nbuf) if gbufs[j] is got.data_pnt), None)
            assert gi == odata_idx, f"frame {i}: data_pnt is buffer {gi}, the oracle's is {odata_idx}"
        if picture is not None:
            assert np.array_equal(gbufs[picture[0]].cpu().numpy(), picture[1]), f"frame {i}: pixels differ"

    for i, (src, key) in enumerate(zip(srcs, keys)):
        if len(inflight) == depth:
            collect()
        if prefetch:
            if drop_ranges_at == i:
                gpu.prefetch(None)
            fetch_range(i // prefetch)
            fetch_range(i // prefetch + 1)
        busy = {k for _, _, k, _, _ in inflight} | {p[0] for _, _, _, _, p in inflight if p is not None}
        oprev = orc.PreviousFrame()
        k = next(j for j in range(nbuf) if obufs[j] is not oprev and j not in busy)
        assert gbufs[k] is not gpu.PreviousFrame()
        raw = bytes(chunks[i])
        if key:
            want = orc.DecompressI(raw, obufs[k])
            t = gpu.DecompressI_async(src, gbufs[k])
        else:
            try:
                odata, osig = orc.DecompressP(raw, obufs[k])
                want = (next((j for j in range(nbuf) if obufs[j] is odata), None), osig)
            except OracleAbort:
                want = "raise"
            t = gpu.DecompressP_async(src, gbufs[k])
        onow = orc.PreviousFrame()
        picture = None
        if onow is not None and want != "raise":
            picture = (next(j for j in range(nbuf) if obufs[j] is onow), onow.copy())
            # adoption is decided by the host stage: PreviousFrame() already answers for the frame just submitted
            assert gpu.PreviousFrame() is gbufs[picture[0]], f"frame {i}: previous frame after submission"
        inflight.append((t, i, k, want, picture))
    while inflight:
        collect()
    if before_close is not None:
        before_close(gpu)
    gpu.StopAndClean()
    if arena is not None:
        arena.close()


@pytest.mark.parametrize("pinned", [False, True], ids=["pageable", "pinned"])
@pytest.mark.parametrize("bits,size", [(16, (320, 240)), (8, (320, 240)), (16, (1920, 1080)), (16, (64, 48))],
                         ids=["16-320x240", "8-320x240", "16-1080p", "16-64x48"])
def test_msvideo1_async_matches_oracle(bits, size, pinned):
    w, h = size
    n = 10 if w * h > 500000 else 24
    frames, keys, pal = sg.msv1_clip(51, w, h, n, bits=bits, p_mix=sg.msv1_p_mix(0.7, 20.0), key_every=9)
    gpu = MSVideo1_16bit(w, h) if bits == 16 else MSVideo1_8bit(w, h, pal)
    gpu.set_option("msv1_parse", "gpu")
    drive(gpu, OracleMSVideo1(bits, w, h, pal), w, h, frames, keys, pinned=pinned)


@pytest.mark.parametrize("bits,size,per_range", [(16, (320, 240), 3), (8, (320, 240), 2), (16, (1920, 1080), 4), (16, (66, 50), 5), (16, (1920, 1088), 3)],
                         ids=["16-320x240", "8-320x240", "16-1080p", "16-66x50", "16-1088p-eight-colour-two-launches"])
def test_msvideo1_async_frames_out_of_prefetched_ranges_match_oracle(bits, size, per_range):
    """jsp_prefetch: the pinned arena the frames lie in (chunk headers between them, as in a file) goes to the device in ranges of a few
    frames, ahead of the frames; frames inside a range queue no upload of their own, a frame a range stops short of takes the ordinary
    way, the ring of four ranges is recycled several times, and halfway every range is given up once.  Same frames, flags and buffer
    identities as the oracle; frames the GPU cannot settle alone (a truncated one, noise) go to the synchronous path as ever."""
    w, h = size
    n = 14 if w * h > 500000 else 32
    big = h == 1088                                              # all-eight-colour key frames of 2.35 MB: past the one-launch form's 2 MiB, scout + decode launches
    frames, keys, pal = sg.msv1_clip(77, w, h, n, bits=bits, p_mix=sg.msv1_p_mix(0.7, 20.0), key_every=2 if big else 9,
                                     key_mix=sg.MIX_ALL_EIGHT if big else sg.MIX_M1)
    if big:
        assert max(len(f) for f in frames) > (2 << 20)
    frames = list(frames)
    frames[5] = frames[5][:len(frames[5]) // 2 + 1]              # ends early, at an odd length
    frames[11] = np.random.default_rng(3).integers(0, 256, 2001, dtype=np.uint8).tobytes()
    gpu = MSVideo1_16bit(w, h) if bits == 16 else MSVideo1_8bit(w, h, pal)
    gpu.set_option("msv1_parse", "gpu")
    seen = {}

    def before_close(g):
        seen["prefetched"] = g.counter("prefetched_frames")
    drive(gpu, OracleMSVideo1(bits, w, h, pal), w, h, frames, keys, pinned=True, prefetch=per_range, drop_ranges_at=n // 2, before_close=before_close)
    # most frames found their bytes on the device: not the ones a range stopped short of, the ones staged synchronously, and the
    # rest of the range that was current when every range was given up
    if w % 4 == 0 and h % 4 == 0:
        assert seen["prefetched"] >= n // 2, seen
    else:
        assert seen["prefetched"] == 0, seen                    # (odd geometry: every frame is staged synchronously, from the caller's bytes)


def test_a_host_address_prefetched_again_means_the_bytes_that_are_there_now():
    """A caller's buffer is a ring: the same pinned address is handed to jsp_prefetch again with OTHER bytes in it (the next stretch of the
    file).  Frames submitted afterwards must decode what is there now — the newest copy of that address — not an older copy still held in
    the codec's ring of ranges (until round 4 the oldest matching range was taken)."""
    import torch
    w, h, n = 320, 240, 6
    clips = [sg.msv1_clip(200 + k, w, h, n, bits=16, key_every=1) for k in range(3)]
    size = max(sum(len(f) + 2 for f in c[0]) for c in clips) + 64
    arena = HostBuffer(size)
    gpu = MSVideo1_16bit(w, h)
    gpu.set_option("msv1_parse", "gpu")
    gpu.Preinit(36)
    gpu.set_option("async_depth", "4")
    bufs = [torch.zeros(w * h, dtype=torch.int32, device="cuda") for _ in range(n)]
    for frames, keys, pal in clips:                              # three passes, three different clips through the SAME host bytes
        pos, srcs = 0, []
        for f in frames:
            arena.array[pos:pos + len(f)] = np.frombuffer(f, dtype=np.uint8)
            srcs.append(arena.array[pos:pos + len(f)])
            pos += len(f) + (len(f) & 1)
        gpu.prefetch(arena.array[:pos])
        orc = OracleMSVideo1(16, w, h, pal)
        orc.Preinit(36)
        tickets = []
        for i, src in enumerate(srcs):
            if len(tickets) == 4:
                gpu.wait(tickets.pop(0))
            tickets.append(gpu.DecompressI_async(src, bufs[i]))
        for t in tickets:
            gpu.wait(t)
        for i, f in enumerate(frames):
            want = np.zeros(w * h, dtype=np.int32)
            assert orc.DecompressI(bytes(f), want) == 0
            assert np.array_equal(bufs[i].cpu().numpy(), want), f"frame {i}: decoded from a stale copy of the range"
    assert gpu.counter("prefetched_frames") == 3 * n
    gpu.StopAndClean()
    arena.close()


@pytest.mark.parametrize("bits,size", [(16, (320, 240)), (8, (320, 240)), (16, (1920, 1080))], ids=["16-320x240", "8-320x240", "16-1080p"])
@pytest.mark.parametrize("pairs", ["on", "off"])
def test_msvideo1_async_two_frames_per_launch(bits, size, pairs):
    """Option msv1_async_pairs (default on): a one-launch frame is held until half of what may be in flight is submitted (at most four frames)
    and they go out in ONE launch — all parse side by side, each paints when the one in front is through (it copies from its pixels and is
    compared with them), and is left unpainted when that one was vetoed.  Key and inter frames, a frame that ends early and a frame of noise
    in the middle (vetoed: re-run by the host together with whatever rode along), depths 4, 5 (two per launch, a frame left over goes out
    alone when it is waited for) and 8 (four per launch).  Same results as the oracle either way; the counter says whether groups were formed."""
    w, h = size
    n = 12 if w * h > 500000 else 30
    frames, keys, pal = sg.msv1_clip(81, w, h, n, bits=bits, p_mix=sg.msv1_p_mix(0.7, 20.0), key_every=7)
    frames = list(frames)
    frames[4] = frames[4][:len(frames[4]) // 3]
    frames[9] = np.random.default_rng(9).integers(0, 256, 1501, dtype=np.uint8).tobytes()
    for depth in (4, 5, 8):
        gpu = MSVideo1_16bit(w, h) if bits == 16 else MSVideo1_8bit(w, h, pal)
        gpu.set_option("msv1_parse", "gpu")
        gpu.set_option("msv1_async_pairs", pairs)
        seen = {}
        drive(gpu, OracleMSVideo1(bits, w, h, pal), w, h, frames, keys, depth=depth, pinned=True,
              before_close=lambda g: seen.update(n=g.counter("paired_frames")))
        if pairs == "on":
            assert seen["n"] >= n // 3, seen
        else:
            assert seen["n"] == 0, seen


def test_prefetched_ranges_recycled_under_frames_held_for_a_group_launch():
    """Eight frames in flight (four to a launch) out of ranges of one or two frames each: the codec keeps four ranges, so the range a held frame reads is
    due to be given up by a prefetch a few frames later — the held frames' kernels must go out before it is (the fuzz campaign found this one:
    wrong pixels, then a fault on a freed range)."""
    w, h, n = 320, 240, 40
    rng = np.random.default_rng(12)
    frames, keys, _ = sg.msv1_clip(83, w, h, n, p_mix=sg.msv1_p_mix(0.6, 15.0), key_every=4)
    frames = [f + bytes(int(rng.integers(0, 4000)) * 2) if k else f for f, k in zip(frames, keys)]   # (sizes vary: the ranges' device copies are reallocated now and then)
    seen = {}
    for per_range in (1, 2):
        gpu = MSVideo1_16bit(w, h)
        gpu.set_option("msv1_parse", "gpu")
        drive(gpu, OracleMSVideo1(16, w, h), w, h, frames, keys, depth=8, pinned=True, prefetch=per_range,
              before_close=lambda g: seen.update(p=g.counter("prefetched_frames"), g=g.counter("paired_frames")))
        assert seen["p"] >= n // 2, seen                       # (every prefetch sends the held frames out first: with a range per frame few groups form)


def test_prefetched_ranges_of_pageable_memory():
    """The file need not be in pinned memory: ranges of an ordinary allocation go up through the runtime's staging, frames out of them
    are decoded from the device copy all the same."""
    w, h, n = 320, 240, 12
    frames, keys, _ = sg.msv1_clip(79, w, h, n, p_mix=sg.msv1_p_mix(0.6, 15.0), key_every=5)
    gpu = MSVideo1_16bit(w, h)
    gpu.set_option("msv1_parse", "gpu")
    seen = {}
    drive(gpu, OracleMSVideo1(16, w, h), w, h, frames, keys, pinned="pageable-arena", prefetch=4,
          before_close=lambda g: seen.update(n=g.counter("prefetched_frames")))
    assert seen["n"] >= n - 2, seen                            # (all but the frame a range stopped short of)


def test_prefetch_is_accepted_and_ignored_where_it_does_not_apply():
    w, h = 64, 48
    frames, keys, _ = sg.msv1_clip(78, w, h, 6, p_mix=sg.msv1_p_mix(0.5, 10.0), key_every=4)
    gpu = MSVideo1_16bit(w, h)
    gpu.set_option("msv1_parse", "host")
    drive(gpu, OracleMSVideo1(16, w, h), w, h, frames, keys, pinned=True, prefetch=2)
    sp = ScreenPressor(64, 48, 24)
    sp.prefetch(np.zeros(100, dtype=np.uint8))
    sp.prefetch(None)
    sp.StopAndClean()


@pytest.mark.parametrize("form", ["auto", "one_launch_dma", "one_launch", "two_launches"])
@pytest.mark.parametrize("pinned", [False, True], ids=["pageable", "pinned"])
def test_msvideo1_async_forms_match_oracle(form, pinned):
    """The three ways a frame runs on the asynchronous path (option msv1_async): one launch fed by the copy engine
    (default), one launch reading the caller's pinned memory itself, scout + decode launches."""
    w, h = 1920, 1080
    frames, keys, _ = sg.msv1_clip(58, w, h, 8, p_mix=sg.msv1_p_mix(0.7, 20.0), key_every=5)
    frames = list(frames)
    frames[3] = frames[3][:len(frames[3]) - 7]                    # ends inside a code, at an odd length
    gpu = MSVideo1_16bit(w, h)
    gpu.set_option("msv1_parse", "gpu")
    gpu.set_option("msv1_async", form)
    drive(gpu, OracleMSVideo1(16, w, h), w, h, frames, keys, depth=4, pinned=pinned)


@pytest.mark.parametrize("depth", [1, 2, 8])
def test_msvideo1_async_hands_unsettled_frames_to_the_synchronous_path(depth):
    """Truncated streams, an all-skip frame, a skip count of zero ("the rest"), random bytes — in the middle of a clip,
    with later frames already in flight behind them."""
    w, h = 320, 240
    frames, keys, _ = sg.msv1_clip(52, w, h, 14, p_mix=sg.msv1_p_mix(0.6, 10.0))
    rng = np.random.default_rng(5)
    frames[3] = frames[3][:len(frames[3]) // 2]                  # stream ends early
    frames[5] = bytes([0x20, 0x84] * 200)                         # skip codes only, longer than size_of_just_skips
    frames[7] = bytes([0x00, 0x84]) + frames[7][2:]              # "skip -1": everything after is copied
    frames[9] = rng.integers(0, 256, 3001, dtype=np.uint8).tobytes()   # noise, odd length
    frames[11] = frames[11][:17]                                  # a few codes only
    gpu = MSVideo1_16bit(w, h)
    gpu.set_option("msv1_parse", "gpu")
    drive(gpu, OracleMSVideo1(16, w, h), w, h, frames, keys, depth=depth)


@pytest.mark.parametrize("form", ["one_launch_dma", "one_launch"])
def test_msvideo1_async_verdict_that_times_out_goes_to_the_synchronous_path(form):
    """One-launch form: a frame whose tiles do not all report in time (GPU shared with other work) gets ONE verdict — the
    time-out's veto, obeyed by every tile —, no pixel of it is written by the launch, and the host re-runs it and the
    frames in flight behind it; the caller sees the same results, later.  (The time-out cannot be provoked on demand:
    option msv1_inject_fault = 2 makes the next such launch deaf to its tiles' reports.)"""
    w, h = 1920, 1080
    frames, keys, _ = sg.msv1_clip(61, w, h, 8, p_mix=sg.msv1_p_mix(0.7, 20.0), key_every=5)
    gpu = MSVideo1_16bit(w, h)
    gpu.set_option("msv1_parse", "gpu")
    gpu.set_option("msv1_async", form)
    gpu.set_option("msv1_inject_fault", "2")
    assert gpu.counter("async_reruns") == 0

    def check(g):
        assert 1 <= g.counter("async_reruns") <= 4, g.counter("async_reruns")   # the deaf frame + what was in flight behind it
        assert g.counter("lookback_fallbacks") == 0
        with pytest.raises(CodecError):
            g.counter("no_such_counter")
    drive(gpu, OracleMSVideo1(16, w, h), w, h, frames, keys, depth=4, before_close=check)


@pytest.mark.parametrize("size", [(320, 240), (1224, 752)], ids=["320x240", "1224x752"])
def test_msvideo1_async_synchronously_staged_frame_waits_for_open_verdicts(size):
    """Found by tests/fuzz_msvideo1.py: a truncated inter frame (its verdict is open until the scout has run), then a
    tiny frame that can only be staged synchronously, handed the buffer the truncated frame's re-run still reads as its
    previous frame (the caller may: it is not PreviousFrame() any more).  The tiny frame must not run ahead."""
    w, h = size
    frames, keys, _ = sg.msv1_clip(57, w, h, 4, p_mix=sg.msv1_p_mix(0.7, 8.0), key_every=0)
    rng = np.random.default_rng(6)
    frames = [frames[0], frames[1][:len(frames[1]) * 3 // 4], rng.integers(0, 256, 92, dtype=np.uint8).tobytes(), frames[3]]
    gpu = MSVideo1_16bit(w, h)
    gpu.set_option("msv1_parse", "gpu")
    drive(gpu, OracleMSVideo1(16, w, h), w, h, frames, [True, False, False, False], depth=2, lines=57)


def test_msvideo1_8bit_end_marker_and_first_frame_skip_code():
    w, h = 64, 32
    frames, keys, pal = sg.msv1_clip(53, w, h, 6, bits=8, p_mix=sg.msv1_p_mix(0.5, 6.0))
    cut = frames[2][:40] + b"\x00\x00" + frames[2][42:]          # end-of-data marker mid frame
    gpu = MSVideo1_8bit(w, h, pal)
    gpu.set_option("msv1_parse", "gpu")
    drive(gpu, OracleMSVideo1(8, w, h, pal), w, h, [frames[0], frames[1], cut, frames[3]], [True, False, False, False], lines=4)
    # a skip code before anything was decoded: the reference raises out of DecompressP, wait() raises too
    gpu = MSVideo1_16bit(w, h)
    gpu.set_option("msv1_parse", "gpu")
    f16, _, _ = sg.msv1_clip(54, w, h, 3, p_mix=sg.msv1_p_mix(0.5, 6.0))
    drive(gpu, OracleMSVideo1(16, w, h), w, h, [f16[1], f16[0], f16[2]], [False, True, False])


@pytest.mark.parametrize("version", [2, 4])
def test_screenpressor_async_matches_oracle(version):
    w, h = 320, 240
    chunks, keys, _ = sg.sp_clip(55, w, h, 14, version=version, key_every=6, flat_at=(8,), unchanged_at=(3,))
    drive(ScreenPressor(w, h, 24), OracleScreenPressor(w, h, 24), w, h, chunks, keys)


@pytest.mark.parametrize("version", [2, 3, 4])
def test_screenpressor_async_groups_of_pictures_on_worker_threads(version):
    """Every coded key frame opens a group of pictures that the asynchronous calls hand to a worker thread and a decoder of
    its own (option sp_async_threads): key frames only, key frames with inter frames behind them, flat key frames (which
    renew nothing and stay in the group in hand), unchanged frames — eight frames in flight, results as the oracle's."""
    w, h = 320, 240
    for seed, n, every, flat, unchanged in ((61, 12, 1, (), ()), (62, 20, 3, (6, 7), (4, 11)), (63, 10, 1, (2, 5), ())):
        chunks, keys, _ = sg.sp_clip(seed, w, h, n, version=version, key_every=every, flat_at=flat, unchanged_at=unchanged)
        gpu = ScreenPressor(w, h, 24)
        gpu.set_option("sp_async_threads", "4")
        drive(gpu, OracleScreenPressor(w, h, 24), w, h, chunks, keys, depth=8)


def _sync_and_async_agree(chunks, keys, w, h, must_show=None):
    """The same frames through the synchronous calls and through the asynchronous ones (worker threads, 8 in flight), a buffer
    per frame: same states, same errors, same previous-frame identities, same pixels in EVERY buffer."""
    import torch
    nbuf = len(chunks)
    results = []
    for mode in ("sync", "async"):
        gpu = ScreenPressor(w, h, 24)
        gpu.Preinit(36)
        gpu.set_option("sp_async_threads", "4")
        gpu.set_option("async_depth", "8")
        bufs = [torch.full((w * h,), 7, dtype=torch.int32, device="cuda") for _ in range(nbuf)]
        log, tickets = [], []

        def settle(i, fn):
            try:
                r = fn()
                out = ("state", int(r)) if keys[i] else ("p", next((k for k in range(nbuf) if bufs[k] is r.data_pnt), None), r.significant_changes)
            except CodecError:
                out = ("raise",)
            log.append((i, out))

        for i, (c, k) in enumerate(zip(chunks, keys)):
            dst = bufs[i]                          # a buffer per frame: no reuse questions
            if mode == "sync":
                settle(i, (lambda: gpu.DecompressI(c, dst)) if k else (lambda: gpu.DecompressP(c, dst)))
            else:
                if len(tickets) == 8:
                    j, t = tickets.pop(0)
                    settle(j, lambda: gpu.wait(t))
                tickets.append((i, (gpu.DecompressI_async if k else gpu.DecompressP_async)(c, dst)))
        for j, t in tickets:
            settle(j, lambda: gpu.wait(t))
        torch.cuda.synchronize()
        prev = gpu.PreviousFrame()
        results.append((log, next((k for k in range(nbuf) if bufs[k] is prev), None), [b.cpu().numpy() for b in bufs]))
        gpu.StopAndClean()
    (log_s, prev_s, pix_s), (log_a, prev_a, pix_a) = results
    assert log_s == log_a
    if must_show:
        assert any(must_show(o) for _, o in log_s), "the broken frame must show"
    assert prev_s == prev_a
    for i, (a, b) in enumerate(zip(pix_s, pix_a)):
        assert np.array_equal(a, b), f"buffer {i}"
    return log_s


def test_screenpressor_async_broken_key_frame_lets_older_state_show_through():
    """A coded key frame that does not decode (cut to a third of its bytes) leaves the decoder the stream had — its
    "a key frame has been decoded" state included — to the frames behind it: the worker that took the broken frame's group,
    on a decoder of its own, must go on with the decoder of the group before.  Such a stream is outside what the oracle
    defines (the reference raises); the bar here is the product's own one-frame-at-a-time path: same states, same errors,
    same previous-frame identities, same pixels."""
    w, h = 320, 240
    chunks, keys, _ = sg.sp_clip(64, w, h, 20, version=4, key_every=4)
    chunks = list(chunks)
    chunks[12] = chunks[12][:len(chunks[12]) // 3]
    _sync_and_async_agree(chunks, keys, w, h, must_show=lambda o: o == ("state", 2) or o == ("raise",))


@pytest.mark.parametrize("version", [2, 4])
def test_screenpressor_async_frame_that_adopts_nothing_against_its_first_byte(version):
    """The asynchronous calls PREDICT from a frame's first byte what it does to the previous frame (ScreenPressor.hx:130-159,
    308-313).  Where the prediction is wrong — an inter frame cut short aborts and adopts nothing; an inter frame behind a
    flat-only start finds no entropy coder and changes nothing — the frames behind it in the same group of pictures must
    still be decoded against the picture that is REALLY there (what the synchronous path copies from), not against the
    failed frame's destination: jsp_wait promises exactly what the synchronous call would have returned, pixels included."""
    w, h = 320, 240
    # (a) an inter frame in the middle of a group, cut to a third of its bytes; three more inter frames behind it
    chunks, keys, _ = sg.sp_clip(66, w, h, 16, version=version, key_every=8)
    chunks = list(chunks)
    assert not keys[3]
    chunks[3] = chunks[3][:max(2, len(chunks[3]) // 3)]
    log = _sync_and_async_agree(chunks, keys, w, h)
    assert log[3][1] == ("raise",) or (log[3][1][0] == "p" and log[3][1][1] != 3), "the cut frame must not have been adopted"
    # (b) a stream that opens with a flat key frame: inter frames behind it have no entropy coder yet
    chunks, keys, _ = sg.sp_clip(67, w, h, 12, version=version, key_every=5)
    chunks = [bytes([((version - 1) << 4) | 1, 10, 20, 30])] + list(chunks[1:])     # frame 0: a flat key frame (head, B, G, R)
    log = _sync_and_async_agree(chunks, keys, w, h)
    assert all(o == ("raise",) or (o[0] == "p" and o[1] != i) for i, o in log if 0 < i < 5), "inter frames behind a flat start adopt nothing"


def test_screenpressor_async_only_caller_keeps_no_history():
    """A caller that only ever uses the asynchronous calls never drains the worker path: the record of finished groups of
    pictures (one per coded key frame) must not grow with the stream."""
    import torch
    w, h = 32, 32
    chunks, keys, _ = sg.sp_clip(68, w, h, 8, version=4, key_every=1)
    gpu = ScreenPressor(w, h, 24)
    gpu.Preinit(36)
    gpu.set_option("sp_async_threads", "4")
    gpu.set_option("async_depth", "8")
    bufs = [torch.zeros(w * h, dtype=torch.int32, device="cuda") for _ in range(24)]
    tickets, worst = [], 0
    for i in range(6000):
        if len(tickets) == 8:
            assert gpu.wait(tickets.pop(0)) == DecoderState.zero_state
        tickets.append(gpu.DecompressI_async(chunks[i % 8], bufs[i % 24]))
        if i % 500 == 499:
            worst = max(worst, gpu.counter("sp_groups_held"))
    for t in tickets:
        gpu.wait(t)
    assert 0 < worst <= 16, f"{worst} groups of pictures on record after thousands of key frames"
    assert gpu.counter("sp_spare_decoders") <= 16
    gpu.StopAndClean()


def test_screenpressor_async_and_synchronous_calls_alternate():
    """The stream's decoder state moves between the worker threads' decoders and the codec's own: synchronous calls after
    asynchronous ones (and back) see one continuous stream."""
    import torch
    w, h = 320, 240
    chunks, keys, _ = sg.sp_clip(65, w, h, 16, version=3, key_every=3)
    gpu, orc = ScreenPressor(w, h, 24), OracleScreenPressor(w, h, 24)
    gpu.Preinit(36)
    orc.Preinit(36)
    gpu.set_option("sp_async_threads", "4")
    gpu.set_option("async_depth", "4")
    gbufs = [torch.zeros(w * h, dtype=torch.int32, device="cuda") for _ in range(8)]
    obufs = [np.zeros(w * h, dtype=np.int32) for _ in range(8)]
    i = 0
    while i < len(chunks):
        if (i // 4) % 2 == 0:                     # four frames through the asynchronous calls ...
            tickets = []
            for j in range(i, min(i + 4, len(chunks))):
                k = next(b for b in range(8) if obufs[b] is not orc.PreviousFrame() and gbufs[b] is not gpu.PreviousFrame() and b not in [t[1] for t in tickets])
                (orc.DecompressI if keys[j] else orc.DecompressP)(chunks[j], obufs[k])
                tickets.append(((gpu.DecompressI_async if keys[j] else gpu.DecompressP_async)(chunks[j], gbufs[k]), k))
            for t, _ in tickets:
                gpu.wait(t)
            i = min(i + 4, len(chunks))
        else:                                     # ... then four through the synchronous ones
            for j in range(i, min(i + 4, len(chunks))):
                k = next(b for b in range(8) if obufs[b] is not orc.PreviousFrame() and gbufs[b] is not gpu.PreviousFrame())
                (orc.DecompressI if keys[j] else orc.DecompressP)(chunks[j], obufs[k])
                (gpu.DecompressI if keys[j] else gpu.DecompressP)(chunks[j], gbufs[k])
            i = min(i + 4, len(chunks))
        assert np.array_equal(gpu.PreviousFrame().cpu().numpy(), orc.PreviousFrame()), f"after frame {i - 1}"
    gpu.StopAndClean()


def test_async_usage_errors():
    import torch
    w, h = 64, 48
    frames, keys, _ = sg.msv1_clip(56, w, h, 4)
    gpu = MSVideo1_16bit(w, h)
    gpu.Preinit(36)
    gpu.set_option("async_depth", "2")
    bufs = [torch.zeros(w * h, dtype=torch.int32, device="cuda") for _ in range(4)]
    t0 = gpu.DecompressI_async(frames[0], bufs[0])
    t1 = gpu.DecompressI_async(frames[1], bufs[1])
    with pytest.raises(CodecError):
        gpu.DecompressI_async(frames[2], bufs[2])               # depth reached
    with pytest.raises(CodecError):
        gpu.set_option("async_depth", "4")                      # not while frames are in flight
    assert gpu.wait(t0) == DecoderState.zero_state
    with pytest.raises(CodecError):
        gpu.wait(t0)                                            # already collected
    assert gpu.wait(t1) == DecoderState.zero_state
    with pytest.raises(CodecError):
        gpu.DecompressI_async(frames[0], np.zeros(w * h, dtype=np.int32))   # host frame buffers are for the synchronous calls
    gpu.StopAndClean()


@pytest.mark.gpu
def test_codec_destroyed_with_frames_in_flight():
    """StopAndClean with tickets never waited for: the kernels of the frames in flight still write the codec's own buffers
    (the report in pinned memory, the stream copy in HBM); destruction waits for them first."""
    import torch
    w, h = 1920, 1080
    frames, keys, _ = sg.msv1_clip(59, w, h, 6)
    for _ in range(3):
        gpu = MSVideo1_16bit(w, h)
        gpu.Preinit(36)
        gpu.set_option("async_depth", "8")
        bufs = [torch.zeros(w * h, dtype=torch.int32, device="cuda") for _ in range(6)]
        for f, b in zip(frames, bufs):
            gpu.DecompressI_async(f, b)
        gpu.StopAndClean()
    torch.cuda.synchronize()
