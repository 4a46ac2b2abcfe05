"""GPU parity tests for ScreenPressor (run with -m gpu): host entropy stage + HIP reconstruction,
through the C ABI, against the CPU oracle on identical bytes — bit-exact frames, identical
significant_changes, identical previous-frame identity."""
import numpy as np
import pytest

from jsplayer_amd import CodecError, DecoderState, ScreenPressor
from jsplayer_amd import streamgen as sg
from oracle_binding import OracleAbort, OracleScreenPressor

pytestmark = pytest.mark.gpu


def dev_buf(n, fill=0, misalign=False):
    """A device frame buffer; `misalign`: a view that starts 4 bytes into its allocation (not 16-byte aligned)."""
    import torch
    if misalign:
        return torch.full((n + 4,), fill, dtype=torch.int32, device="cuda")[1:1 + n]
    return torch.full((n,), fill, dtype=torch.int32, device="cuda")


def to_np(t):
    return t.cpu().numpy() if hasattr(t, "cpu") else t


def drive_pair(w, h, bpp, chunks, keys, frames=None, lines=36, nbuf=3, host=False, band_rows=None, misalign=False):
    orc, gpu = OracleScreenPressor(w, h, bpp), ScreenPressor(w, h, bpp)
    orc.Preinit(lines)
    gpu.Preinit(lines)
    if band_rows is not None:
        gpu.set_option("sp_band_rows", str(band_rows))
    obufs = [np.full(w * h, 0x00A5A5A5, dtype=np.int32) for _ in range(nbuf)]
    gbufs = [np.full(w * h, 0x00A5A5A5, dtype=np.int32) if host else dev_buf(w * h, 0x00A5A5A5, misalign) for _ in range(nbuf)]
    for i, (src, key) in enumerate(zip(chunks, keys)):
        oprev, gprev = orc.PreviousFrame(), gpu.PreviousFrame()
        oi = next(k for k in range(nbuf) if obufs[k] is not oprev)
        gi = next(k for k in range(nbuf) if gbufs[k] is not gprev)
        assert oi == gi
        assert gpu.IsKeyFrame(src) == orc.IsKeyFrame(src)
        if key:
            rc = orc.DecompressI(src, obufs[oi])
            st = gpu.DecompressI(src, gbufs[gi])
            assert (rc == 0) == (st == DecoderState.zero_state), f"frame {i}: oracle {rc} vs {st}"
        else:
            try:
                odata, osig = orc.DecompressP(src, obufs[oi])
            except OracleAbort:
                with pytest.raises(CodecError):
                    gpu.DecompressP(src, gbufs[gi])
            else:
                res = gpu.DecompressP(src, gbufs[gi])
                assert res.significant_changes == osig, f"frame {i}"
                assert (res.data_pnt is gbufs[gi]) == (odata is obufs[oi]), f"frame {i}"
                assert (res.data_pnt is None) == (odata is None)
        onow, gnow = orc.PreviousFrame(), gpu.PreviousFrame()
        assert (onow is None) == (gnow is None), f"frame {i}"
        if onow is not None:
            assert [k for k in range(nbuf) if obufs[k] is onow] == [k for k in range(nbuf) if gbufs[k] is gnow]
            assert np.array_equal(onow, to_np(gnow)), f"frame {i}: previous frame differs"
            if frames is not None:
                assert np.array_equal(onow.view(np.uint32), frames[i]), f"frame {i}: not the encoded image"
    gpu.StopAndClean()


SIZES = [(64, 48), (320, 240), (100, 52), (37, 23), (1920, 1080)]


@pytest.mark.parametrize("version", [2, 3, 4])
@pytest.mark.parametrize("size", SIZES, ids=[f"{w}x{h}" for w, h in SIZES])
def test_clip_parity_device(version, size):
    w, h = size
    n = 4 if w * h > 500000 else 12
    chunks, keys, frames = sg.sp_clip(900 + version, w, h, n, version=version, key_every=7, flat_at=(5,),
                                      unchanged_at=(2,))
    drive_pair(w, h, 24, chunks, keys, frames)


@pytest.mark.parametrize("band_rows", [0, 1, 3, 16, 25, "auto"])
@pytest.mark.parametrize("size", [(64, 48), (100, 52), (37, 23), (320, 240), (2052, 40), (5000, 20)],
                         ids=lambda s: f"{s[0]}x{s[1]}")
def test_key_frame_bands(size, band_rows):
    """Key frames rebuilt band by band (one workgroup each, started from the host stage's seed rows) give
    the same pictures whatever the band height: both row kernels, flat frames, frames after inter frames."""
    w, h = size
    chunks, keys, frames = sg.sp_clip(990, w, h, 7, version=4, key_every=3, flat_at=(4,), rects=40, gradients=10)
    drive_pair(w, h, 24, chunks, keys, frames, band_rows=band_rows)


@pytest.mark.parametrize("size", [(64, 48), (320, 240), (100, 52)], ids=lambda s: f"{s[0]}x{s[1]}")
def test_frame_buffers_that_are_not_16_byte_aligned(size):
    """Caller buffers only have to be int32 arrays: a view 4 bytes into an allocation routes key frames to the
    row-major layout and the per-lane-search kernel, inter frames to the scalar-store paths — per call and as a
    staged batch with fused inter frames."""
    w, h = size
    chunks, keys, frames = sg.sp_clip(993, w, h, 9, version=4, key_every=5, unchanged_at=(2,))
    drive_pair(w, h, 24, chunks, keys, frames, misalign=True)
    gpu = ScreenPressor(w, h, 24)
    gpu.Preinit(36)
    dsts = [dev_buf(w * h, -1, misalign=True) for _ in range(9)]
    assert all(d.data_ptr() % 16 == 4 for d in dsts)
    st = gpu.stage_batch(chunks, dsts, is_key=keys)
    st.decode()
    gpu.sync()
    _, adopted, _ = st.results()
    for i, (d, img) in enumerate(zip(dsts, frames)):
        if adopted[i]:
            assert np.array_equal(to_np(d).view(np.uint32), img), f"frame {i}"
    st.close()


def test_noise_key_frame_rows_with_more_runs_than_the_tile_window():
    """Almost every pixel its own run: a 512-column span row then has more records than a tile's LDS window
    holds, and the kernel scatters that row straight from global memory."""
    w, h = 1024, 40
    chunks, keys, frames = sg.sp_clip(992, w, h, 2, version=4, key_every=1, noise=0.97)
    drive_pair(w, h, 24, chunks, keys, frames)
    drive_pair(w, h, 24, chunks, keys, frames, band_rows=7)


def test_noise_key_frame_in_one_tall_band_takes_the_direct_scatter():
    """One band of 300 rows: the tile's row index and left pixels leave its LDS slice a window of 128 records, a noisy
    256-column row has up to 256 — every such row is scattered straight from global memory (the `direct` branch), the
    quiet rows in between go through 128-record windows; and a workgroup of such tall tiles has fewer than eight waves."""
    w, h = 1024, 300
    chunks, keys, frames = sg.sp_clip(993, w, h, 2, version=4, key_every=1, noise=0.97)
    drive_pair(w, h, 24, chunks, keys, frames, band_rows=0)
    chunks, keys, frames = sg.sp_clip(994, w, h, 2, version=4, key_every=1, noise=0.3)
    drive_pair(w, h, 24, chunks, keys, frames, band_rows=0)


def test_frames_taller_than_a_tiles_index_are_cut_into_bands():
    """4 200 rows: more than one wave's LDS slice can index (4 096), so 'one band per frame' is cut at 4 096; a workgroup is one wave."""
    w, h = 64, 4200
    chunks, keys, frames = sg.sp_clip(995, w, h, 2, version=4, key_every=1)
    drive_pair(w, h, 24, chunks, keys, frames, band_rows=0)
    drive_pair(w, h, 24, chunks, keys, frames)


def test_key_frame_bands_1080p_batch():
    """A batch of 1080p key frames in one launch, banded automatically, with an odd band height, in bands too tall for eight
    waves to share a workgroup's LDS (600 rows) and as one band per frame (1 080 rows: four waves per workgroup)."""
    w, h = 1920, 1080
    chunks, keys, frames = sg.sp_clip(991, w, h, 3, version=4, key_every=1)
    for rows in ("auto", "37", "600", "0"):
        gpu = ScreenPressor(w, h, 24)
        gpu.Preinit(36)
        gpu.set_option("sp_band_rows", rows)
        dsts = [dev_buf(w * h, -1) for _ in range(3)]
        st = gpu.stage_batch(chunks, dsts, is_key=keys)
        assert st.info()["kernel_launches"] == 1
        st.decode()
        gpu.sync()
        for d, img in zip(dsts, frames):
            assert np.array_equal(to_np(d).view(np.uint32), img)
        st.close()
        gpu.StopAndClean()
    with pytest.raises(CodecError):
        ScreenPressor(64, 48, 24).set_option("sp_band_rows", "minus one")


@pytest.mark.parametrize("version", [2, 4])
def test_clip_parity_16bpp(version):
    w, h = 160, 96
    chunks, keys, frames = sg.sp_clip(910, w, h, 8, bpp=16, version=version, flat_at=(4,))
    drive_pair(w, h, 16, chunks, keys, frames)


@pytest.mark.parametrize("version", [2, 4])
def test_clip_parity_host_pointers(version):
    w, h = 320, 240
    chunks, keys, frames = sg.sp_clip(920, w, h, 8, version=version, key_every=5)
    drive_pair(w, h, 24, chunks, keys, frames, host=True)


def test_bad_headers_and_flat_first_frame():
    w, h = 32, 32
    chunks, keys, _ = sg.sp_clip(930, w, h, 2, version=4)
    bad = [bytes([0x13, 0, 0, 0]), bytes([0x52, 0, 0, 0, 0, 0]), bytes([0x11, 1, 2, 3]), b""]
    drive_pair(w, h, 24, bad + chunks, [True] * len(bad) + keys)


@pytest.mark.parametrize("version", [2, 4])
def test_truncated_key_frame(version):
    w, h = 64, 48
    chunks, keys, _ = sg.sp_clip(940, w, h, 3, version=version)
    cut = chunks[0][: len(chunks[0]) // 2]
    # v2: the cut poisons the range coder and the reference never returns -> error, prevFrame null.
    # v3/v4: missing bytes read as 0 and some garbage frame comes out (compared pixel for pixel).
    # Either way the next coded key frame resets every model and decoding recovers.  (Inter frames
    # decoded against a garbage key frame are outside the contract: DESIGN.md "invalid streams".)
    drive_pair(w, h, 24, [chunks[0], chunks[1], cut, chunks[0], chunks[1], chunks[2]],
               [True, False, True, True, False, False])


def test_batch_of_key_frames_one_launch_and_p_clip():
    import torch
    w, h = 640, 360
    # independent I-frames (each from its own encoder): one launch for the whole batch
    streams, images = [], []
    for i in range(6):
        c, k, f = sg.sp_clip(950 + i, w, h, 1, version=4)
        streams.append(c[0])
        images.append(f[0])
    gpu = ScreenPressor(w, h, 24)
    gpu.Preinit(36)
    dsts = [dev_buf(w * h, -1) for _ in range(6)]
    st = gpu.stage_batch(streams, dsts)
    info = st.info()
    assert info["kernel_launches"] == 1 and info["runs"] > 0
    st.decode()
    gpu.sync()
    status, adopted, _ = st.results()
    assert status == [0] * 6 and adopted == [1] * 6
    for d, img in zip(dsts, images):
        assert np.array_equal(to_np(d).view(np.uint32), img)
    st.close()
    # a P clip staged as one batch: one launch per frame, replayable
    chunks, keys, frames = sg.sp_clip(960, w, h, 10, version=4)
    gpu2 = ScreenPressor(w, h, 24)
    gpu2.Preinit(36)
    dsts = [dev_buf(w * h, -1) for _ in range(10)]
    st = gpu2.stage_batch(chunks, dsts, is_key=keys)
    for _ in range(2):
        st.decode()
    gpu2.sync()
    for d, img in zip(dsts, frames):
        assert np.array_equal(to_np(d).view(np.uint32), img)
    info = st.info()
    assert info["kernel_launches"] == 2        # the key frame, then all nine inter frames in one launch
    st.close()
    # the same clip with inter-frame fusion off: one launch per frame, same pictures
    gpu3 = ScreenPressor(w, h, 24)
    gpu3.Preinit(36)
    gpu3.set_option("sp_inter_fusion", "off")
    dsts = [dev_buf(w * h, -1) for _ in range(10)]
    st = gpu3.stage_batch(chunks, dsts, is_key=keys)
    assert st.info()["kernel_launches"] == 10
    st.decode()
    gpu3.sync()
    for d, img in zip(dsts, frames):
        assert np.array_equal(to_np(d).view(np.uint32), img)
    st.close()


@pytest.mark.parametrize("size", [(640, 360), (100, 52), (37, 23), (2052, 40)], ids=lambda s: f"{s[0]}x{s[1]}")
@pytest.mark.parametrize("version", [2, 4])
def test_fused_inter_frames(size, version):
    """A staged clip whose inter frames share launches: heavy-motion frames (more than a quarter of the
    pixels moved) break the run and keep their motion blocks, unchanged frames sit inside it, key frames
    split it, and frame buffers are reused round-robin as a player's pool would."""
    w, h = size
    chunks, keys, frames = sg.sp_clip(970 + version, w, h, 16, version=version, key_every=9, unchanged_at=(3, 11),
                                      p_mix_at={5: dict(unchanged=0.30, motion=0.60), 6: dict(unchanged=0.05, motion=0.90)})
    gpu = ScreenPressor(w, h, 24)
    gpu.Preinit(36)
    # distinct buffers: every frame can be checked
    dsts = [dev_buf(w * h, -1) for _ in range(16)]
    st = gpu.stage_batch(chunks, dsts, is_key=keys)
    launches = st.info()["kernel_launches"]
    for _ in range(2):                       # replayable
        st.decode()
    gpu.sync()
    status, adopted, _ = st.results()
    assert status == [0] * 16
    for i, (d, img) in enumerate(zip(dsts, frames)):
        if adopted[i]:
            assert np.array_equal(to_np(d).view(np.uint32), img), f"frame {i}"
    assert not adopted[3] and not adopted[11]
    assert launches < 12                     # 2 key frames + a few groups/breakers, not one per frame
    st.close()
    gpu.StopAndClean()
    # a pool of three buffers used as a player would (never the buffer holding the previous frame; a frame
    # that changes nothing does not consume one): afterwards each buffer holds the last frame decoded into it
    gpu = ScreenPressor(w, h, 24)
    gpu.Preinit(36)
    pool = [dev_buf(w * h, -1) for _ in range(3)]
    order, k = [], 0
    for i in range(16):
        order.append(pool[k % 3])
        k += 1 if adopted[i] else 0
    st = gpu.stage_batch(chunks, order, is_key=keys)
    st.decode()
    gpu.sync()
    assert st.results()[1] == adopted
    for b in pool:
        mine = [i for i in range(16) if adopted[i] and order[i] is b]
        assert mine and np.array_equal(to_np(b).view(np.uint32), frames[mine[-1]]), f"frame {mine[-1]}"
    st.close()


@pytest.mark.parametrize("size", [(640, 360), (100, 52), (1928, 24)], ids=lambda s: f"{s[0]}x{s[1]}")
def test_inter_groups_at_odd_sizes_and_in_a_three_buffer_rotation(size):
    """Runs of inter frames in one launch (sp_pframe_group_kernel: the workgroups walk the whole group with their pixels in registers):
    every frame of a 75-frame clip against the images the encoder was given — full and sub-rectangle repaints, literalised motion,
    unchanged frames inside the run, the partial last block row and column, replays — with a buffer per frame and with three buffers
    in rotation (what a player has)."""
    w, h = size
    n = 76
    chunks, keys, frames = sg.sp_clip(985, w, h, n, version=4, unchanged_at=(7, 20, 21), p_mix_at={9: dict(unchanged=0.5, motion=0.2), 30: dict(unchanged=0.97, motion=0.01)})
    gpu = ScreenPressor(w, h, 24)
    gpu.Preinit(36)
    dsts = [dev_buf(w * h, -1) for _ in range(n)]
    st = gpu.stage_batch(chunks, dsts, is_key=keys)
    assert "sp_pframe_group_kernel" in st.kernels(), st.kernels()
    for _ in range(2):
        st.decode()
    gpu.sync()
    status, adopted, _ = st.results()
    assert status == [0] * n
    for i, (d, img) in enumerate(zip(dsts, frames)):
        if adopted[i]:
            assert np.array_equal(to_np(d).view(np.uint32), img), f"frame {i}"
    assert not adopted[7] and not adopted[20]
    st.close()
    gpu.StopAndClean()
    gpu = ScreenPressor(w, h, 24)
    gpu.Preinit(36)
    pool = [dev_buf(w * h, -1) for _ in range(3)]
    order, k = [], 0
    for i in range(n):
        order.append(pool[k % 3])
        k += 1 if adopted[i] else 0
    st = gpu.stage_batch(chunks, order, is_key=keys)
    st.decode()
    gpu.sync()
    for b in pool:
        mine = [i for i in range(n) if adopted[i] and order[i] is b]
        assert mine and np.array_equal(to_np(b).view(np.uint32), frames[mine[-1]]), f"frame {mine[-1]}"
    st.close()
    gpu.StopAndClean()


@pytest.mark.parametrize("head", [0x12, 0x22, 0x32])
def test_garbage_streams_never_fault(head):
    """Random bytes behind a valid header: the reference itself raises, spins or paints noise on such
    input ("invalid ScreenPressor streams"), so there is no parity claim — but the host
    stage must hand the kernels only in-range descriptors and the codec must stay usable."""
    w, h = 320, 240
    rng = np.random.default_rng(head)
    gpu = ScreenPressor(w, h, 24)
    gpu.Preinit(36)
    bufs = [dev_buf(w * h, 0) for _ in range(3)]
    good, keys, frames = sg.sp_clip(970, w, h, 2, version=(head >> 4) + 1)
    assert good[1] != b"\x00"
    assert gpu.DecompressI(good[0], bufs[0]) == DecoderState.zero_state
    for i in range(60):
        junk = bytes([head if i % 3 else 1]) + rng.integers(0, 256, size=int(rng.integers(1, 400)), dtype=np.uint8).tobytes()
        dst = next(b for b in bufs if b is not gpu.PreviousFrame())
        if i % 2:
            st = gpu.DecompressI(junk, dst)
            assert st in (DecoderState.zero_state, DecoderState.error_occured)
        else:
            try:
                gpu.DecompressP(junk, dst)
            except CodecError:
                pass
    # a fresh coded key frame resets every model: decoding is exact again
    dst = next(b for b in bufs if b is not gpu.PreviousFrame())
    assert gpu.DecompressI(good[0], dst) == DecoderState.zero_state
    assert np.array_equal(to_np(dst).view(np.uint32), frames[0])
    dst2 = next(b for b in bufs if b is not gpu.PreviousFrame())
    res = gpu.DecompressP(good[1], dst2)
    assert res.data_pnt is dst2 and np.array_equal(to_np(dst2).view(np.uint32), frames[1])


@pytest.mark.parametrize("size", [(2052, 40), (4096, 36), (4104, 33), (5000, 20)], ids=lambda s: f"{s[0]}x{s[1]}")
def test_wide_frames_use_the_other_kernel_instantiations(size):
    """Widths beyond 2048 / 4096 pixels select the 1024-lane and the 8-pixels-per-lane instantiations
    of the row kernel (and 5000 is not a multiple of 16: partial blocks on the right edge)."""
    w, h = size
    chunks, keys, frames = sg.sp_clip(980, w, h, 4, version=4, rects=30, gradients=8)
    drive_pair(w, h, 24, chunks, keys, frames)


@pytest.mark.gpu
def test_batches_staged_into_one_batch_object():
    """jsp_restage_batch (stage_batch(reuse=)): a clip taken in three batches of different sizes through ONE batch object —
    its pinned and device buffers taken over each time, the host stage of every batch with its groups of pictures side by
    side — leaves the frames the encoder was given; also for MSVideo1 against the oracle."""
    w, h = 640, 360
    chunks, keys, frames = sg.sp_clip(994, w, h, 30, version=4, key_every=4, unchanged_at=(5, 17), flat_at=(9,))
    gpu = ScreenPressor(w, h, 24)
    gpu.Preinit(36)
    dsts = [dev_buf(w * h, -1) for _ in range(30)]
    st = None
    for lo, hi in ((0, 7), (7, 23), (23, 30)):
        st = gpu.stage_batch(chunks[lo:hi], dsts[lo:hi], is_key=keys[lo:hi], reuse=st)
        st.decode()
        gpu.sync()
        status, adopted, _ = st.results()
        assert all(s == 0 for s in status)
        for i in range(lo, hi):
            if adopted[i - lo]:
                assert np.array_equal(to_np(dsts[i]).view(np.uint32), frames[i]), f"frame {i}"
    st.close()
    from jsplayer_amd import MSVideo1_16bit
    from oracle_binding import OracleMSVideo1
    mf, mk, _ = sg.msv1_clip(995, w, h, 12, p_mix=sg.msv1_p_mix(0.6, 10.0), key_every=5)
    g16, orc = MSVideo1_16bit(w, h), OracleMSVideo1(16, w, h)
    g16.set_option("msv1_parse", "gpu")
    want = []
    prev = None
    for f, k in zip(mf, mk):
        buf = np.full(w * h, -1, dtype=np.int32) if prev is None else prev.copy()
        if k:
            orc.DecompressI(f, buf)
        else:
            orc.DecompressP(f, buf)
        p = orc.PreviousFrame()
        prev = p if p is not None else buf
        want.append(prev.copy())
    md = [dev_buf(w * h, -1) for _ in range(12)]
    st = None
    for lo, hi in ((0, 5), (5, 12)):
        st = g16.stage_batch(mf[lo:hi], md[lo:hi], is_key=mk[lo:hi], reuse=st)
        st.decode()
        g16.sync()
        _, adopted, _ = st.results()
        for i in range(lo, hi):
            if adopted[i - lo]:
                assert np.array_equal(to_np(md[i]), want[i]), f"msvideo1 frame {i}"
    st.close()


@pytest.mark.parametrize("how", ["sync", "async", "async_workers", "host_pointers", "batch"])
@pytest.mark.parametrize("nbuf", [2, 3, 4])
def test_left_of_column_zero_reads_what_the_callers_buffer_holds(nbuf, how):
    """The one place where an inter frame reads its destination before writing it (tests/test_screenpressor_cpu.py::
    test_left_of_column_zero_is_the_destinations_old_content), through the C ABI: whatever the caller's buffer rotation — two
    buffers (the picture two frames back), three or four (the buffer's fill shows through) — the product hands back what the
    reference hands back, through the synchronous calls, the asynchronous ones (host stage inside the call, or on worker
    threads), host frame buffers and a staged batch.  (Round 3 read the picture two frames back whatever the rotation.)"""
    import torch
    from test_screenpressor_cpu import column0_clip, oracle_with_rotation
    w, h, y0, chunks, frames = column0_clip(4)
    want = oracle_with_rotation(w, h, chunks, nbuf)
    if nbuf > 2:
        assert int(want[y0, 0]) == 0x123456          # the third buffer's fill shows through in the reference
    gpu = ScreenPressor(w, h, 24)
    gpu.Preinit(36)
    if how == "host_pointers":
        bufs = [np.full(w * h, 0x123456, dtype=np.int32) for _ in range(nbuf)]
    else:
        bufs = [torch.full((w * h,), 0x123456, dtype=torch.int32, device="cuda") for _ in range(nbuf)]
    order = [bufs[0], bufs[1], bufs[2 % nbuf]]
    if how in ("sync", "host_pointers"):
        assert gpu.DecompressI(chunks[0], order[0]) == 0
        gpu.DecompressP(chunks[1], order[1])
        gpu.DecompressP(chunks[2], order[2])
    elif how == "batch":
        if nbuf == 2:
            # (the same buffer twice in one batch: staged in order)
            pass
        st = gpu.stage_batch(chunks, order, is_key=[True, False, False])
        st.decode()
        gpu.sync()
        assert st.results()[0] == [0, 0, 0]
        st.close()
    else:
        gpu.set_option("sp_async_threads", "4" if how == "async_workers" else "1")
        tickets = [gpu.DecompressI_async(chunks[0], order[0]), gpu.DecompressP_async(chunks[1], order[1])]
        if nbuf == 2:
            for t in tickets:
                gpu.wait(t)                          # (its destination is the first frame's buffer: not while that is in flight)
            tickets = []
        tickets.append(gpu.DecompressP_async(chunks[2], order[2]))
        for t in tickets:
            gpu.wait(t)
    got = gpu.PreviousFrame()
    got = (got if how == "host_pointers" else got.cpu().numpy()).view(np.uint32).reshape(h, w)
    assert np.array_equal(got, want)
    gpu.StopAndClean()
