"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against
the CPU oracle on identical bytes.  Integer/byte work: the bar is bit-exact frames, identical
significant_changes and identical "which buffer is prevFrame" answers."""
import json
import os

import numpy as np
import pytest

from jsplayer_amd import CodecError, MSVideo1_16bit, MSVideo1_8bit
from jsplayer_amd import streamgen as sg
from oracle_binding import OracleAbort, OracleMSVideo1
from test_msvideo1_oracle import KATS, run_kat

pytestmark = pytest.mark.gpu


def torch_mod():
    import torch
    return torch


PARSE_MODE = "host"


@pytest.fixture(autouse=True, params=["host", "gpu"])
def parse_mode(request):
    """Every test in this file runs twice: descriptor tables from the sequential host parser and
    from the on-GPU parse kernels (msv1_parse_kernels.hip).  Results must be identical."""
    global PARSE_MODE
    PARSE_MODE = request.param
    yield request.param
    PARSE_MODE = "host"


def make_gpu(bits, w, h, pal=None):
    c = MSVideo1_16bit(w, h) if bits == 16 else MSVideo1_8bit(w, h, pal or b"")
    c.set_option("msv1_parse", PARSE_MODE)
    return c


def dev_buf(n, fill=0, misalign=False):
    """A device frame buffer; `misalign`: a view that starts 4 bytes into its allocation (not 16-byte aligned)."""
    torch = torch_mod()
    if misalign:
        return torch.full((n + 4,), fill, dtype=torch.int32, device="cuda")[1:1 + n]
    return torch.full((n,), fill, dtype=torch.int32, device="cuda")


def to_np(t):
    return t.cpu().numpy() if hasattr(t, "cpu") else t


def test_native_library_is_the_one_loaded():
    from jsplayer_amd import _native
    assert os.path.exists(_native.LIB_PATH)
    assert _native.lib().jsp_version().decode().endswith("gfx950")


@pytest.mark.parametrize("kat", KATS, ids=[k["name"] for k in KATS])
def test_known_answers_device_buffers(kat):
    run_kat(make_gpu, dev_buf, to_np, kat)


@pytest.mark.parametrize("kat", KATS, ids=[k["name"] for k in KATS])
def test_known_answers_host_buffers(kat):
    run_kat(make_gpu, lambda n, fill: np.full(n, fill, dtype=np.int32), to_np, kat)


def drive_pair(bits, w, h, frames, keys, pal, lines=36, nbuf=3, host=False, prefill=0x00A5A5A5, misalign=False):
    """Feed the same clip to the oracle and to the HIP path with Manager's buffer protocol and
    compare everything observable after every frame."""
    orc = OracleMSVideo1(bits, w, h, pal)
    gpu = make_gpu(bits, w, h, pal)
    orc.Preinit(lines)
    gpu.Preinit(lines)
    obufs = [np.full(w * h, prefill, dtype=np.int32) for _ in range(nbuf)]
    gbufs = [np.full(w * h, prefill, dtype=np.int32) if host else dev_buf(w * h, prefill, misalign) for _ in range(nbuf)]
    for i, (src, key) in enumerate(zip(frames, keys)):
        oprev, gprev = orc.PreviousFrame(), gpu.PreviousFrame()
        oi = next(k for k in range(nbuf) if obufs[k] is not oprev)
        gi = next(k for k in range(nbuf) if gbufs[k] is not gprev)
        assert oi == gi, "buffer choice diverged"
        assert gpu.IsKeyFrame(src) == orc.IsKeyFrame(src)
        if key:
            assert orc.DecompressI(src, obufs[oi]) == 0
            assert gpu.DecompressI(src, gbufs[gi]) == 0
        else:
            try:
                odata, osig = orc.DecompressP(src, obufs[oi])
            except OracleAbort:
                with pytest.raises(CodecError):
                    gpu.DecompressP(src, gbufs[gi])
                odata = None
            else:
                res = gpu.DecompressP(src, gbufs[gi])
                assert res.significant_changes == osig, f"frame {i}: significant_changes"
                assert (res.data_pnt is gbufs[gi]) == (odata is obufs[oi]), f"frame {i}: data_pnt identity"
                assert (res.data_pnt is None) == (odata is None)
        onow, gnow = orc.PreviousFrame(), gpu.PreviousFrame()
        assert (onow is None) == (gnow is None)
        if onow is not None:
            assert [k for k in range(nbuf) if obufs[k] is onow] == [k for k in range(nbuf) if gbufs[k] is gnow]
        # every buffer, adopted or not, must hold the same pixels (the reference paints in place)
        for k in range(nbuf):
            assert np.array_equal(obufs[k], to_np(gbufs[k])), f"frame {i}: buffer {k} differs"
    gpu.StopAndClean()


SIZES = [(4, 4), (16, 16), (64, 48), (320, 240), (100, 52), (37, 23), (1920, 1080)]


@pytest.mark.parametrize("bits", [16, 8])
@pytest.mark.parametrize("size", SIZES, ids=[f"{w}x{h}" for w, h in SIZES])
def test_clip_parity_device(bits, size):
    w, h = size
    n = 4 if w * h > 500000 else 10
    frames, keys, pal = sg.msv1_clip(20 + bits, w, h, n, bits=bits, p_mix=sg.msv1_p_mix(0.7, 40.0), key_every=5)
    drive_pair(bits, w, h, frames, keys, pal)


@pytest.mark.parametrize("bits", [16, 8])
@pytest.mark.parametrize("size", [(64, 48), (37, 23), (320, 240)], ids=["64x48", "37x23", "320x240"])
def test_clip_parity_host_pointers(bits, size):
    w, h = size
    frames, keys, pal = sg.msv1_clip(40 + bits, w, h, 8, bits=bits, p_mix=sg.msv1_p_mix(0.5, 9.0))
    drive_pair(bits, w, h, frames, keys, pal, host=True)


@pytest.mark.parametrize("bits", [16, 8])
@pytest.mark.parametrize("mix", [sg.MIX_ALL_SOLID, sg.MIX_ALL_EIGHT, sg.Msv1Mix(0.0, 1.0, 0.0)],
                         ids=["solid", "eight", "two"])
def test_extreme_mixes(bits, mix):
    w, h = 320, 240
    frames, keys, pal = sg.msv1_clip(60, w, h, 2, bits=bits, key_mix=mix)
    drive_pair(bits, w, h, frames, keys, pal)


@pytest.mark.parametrize("bits", [16, 8])
def test_truncated_and_garbage_streams(bits):
    """Streams cut at every length + random bytes: the JS out-of-range semantics the oracle
    documents must come out of the kernel too."""
    w, h = 32, 16
    frames, keys, pal = sg.msv1_clip(70 + bits, w, h, 2, bits=bits, p_mix=sg.msv1_p_mix(0.4, 5.0))
    rng = np.random.default_rng(5)
    clips = []
    full = frames[1]
    for cut in list(range(0, 40)) + [len(full) // 2, len(full) - 1]:
        clips.append(full[:cut])
    for _ in range(40):
        clips.append(rng.integers(0, 256, size=int(rng.integers(1, 200)), dtype=np.uint8).tobytes())
    srcs = [frames[0]]
    ks = [True]
    for c in clips:
        srcs.append(c)
        ks.append(False)
    drive_pair(bits, w, h, srcs, ks, pal, lines=4)


def test_skip_before_first_frame_is_an_error():
    w, h = 16, 8
    src = bytes([0x00, 0xFC, 0x01, 0x84] + [0] * 8)
    drive_pair(16, w, h, [src], [False], None, lines=0)


def test_negative_skip_and_all_skip_long_stream():
    w, h = 64, 32
    frames, _, _ = sg.msv1_clip(81, w, h, 1)
    neg = bytes([0x1F, 0x80, 0x00, 0x84] + [0] * 20)
    # 128 blocks, all-skip stream longer than size_of_just_skips: full copy, not adopted
    allskip = bytes([0x10, 0x84] * 8)
    drive_pair(16, w, h, [frames[0], neg, allskip, neg], [True, False, False, False], None, lines=0)


def test_8bit_end_marker_mid_frame():
    w, h = 32, 16
    frames, _, pal = sg.msv1_clip(82, w, h, 2, bits=8)
    cut = frames[1][:20] + b"\x00\x00" + frames[1][22:]
    drive_pair(8, w, h, [frames[0], cut, frames[1]], [True, False, False], pal, lines=4)


def test_batch_matches_sequential_and_full_size_properties():
    """BASELINE config 2 shape: key-frame-only 1920x1080, one launch for the whole batch."""
    torch = torch_mod()
    w, h, n = 1920, 1080, 6
    frames, keys, _ = sg.msv1_clip(2, w, h, n)
    gpu = make_gpu(16, w, h)
    gpu.Preinit(36)
    dsts = [dev_buf(w * h, -1) for _ in range(n)]
    st = gpu.stage_batch(frames, dsts)
    info = st.info()
    assert info["kernel_launches"] == 1     # host tables: the block kernel; raw bytes: the fused parse + block kernel
    assert st.kernels() == ("msv1_blocks_kernel" if PARSE_MODE == "host" else "msv1_fused_kernel")
    assert info["moved_bytes"] == info["algorithmic_bytes"] + (n * 129600 * 4 if PARSE_MODE == "host" else 0)
    assert info["units_coded"] == n * 129600 and info["units_copied"] == 0
    assert info["algorithmic_bytes"] == sum(len(f) for f in frames) + n * w * h * 4
    st.decode()
    gpu.sync()
    status, adopted, _ = st.results()
    assert status == [0] * n and adopted == [1] * n
    assert gpu.PreviousFrame() is dsts[-1]
    # decoding the staged batch again is idempotent
    first = [d.clone() for d in dsts]
    st.decode()
    gpu.sync()
    for a, b in zip(first, dsts):
        assert torch.equal(a, b)
    # frame-by-frame through the oracle
    orc = OracleMSVideo1(16, w, h)
    orc.Preinit(36)
    ref = np.zeros(w * h, dtype=np.int32)
    for src, d in zip(frames, dsts):
        orc.DecompressI(src, ref)
        assert np.array_equal(ref, to_np(d))
    # size-independent property: every pixel is a left-aligned 5-bit-per-channel colour
    assert int((dsts[0] & ~0x00F8F8F8).abs().max()) == 0
    st.close()


def test_batch_with_inter_frames_matches_oracle():
    w, h, n = 320, 240, 12
    frames, keys, _ = sg.msv1_clip(1, w, h, n, p_mix=sg.msv1_p_mix(0.7, 40.0), key_every=6)
    gpu = make_gpu(16, w, h)
    gpu.Preinit(36)
    dsts = [dev_buf(w * h, 3) for _ in range(n)]
    st = gpu.stage_batch(frames, dsts, is_key=keys)
    st.decode()
    gpu.sync()
    status, adopted, signif = st.results()
    orc = OracleMSVideo1(16, w, h)
    orc.Preinit(36)
    obufs = [np.full(w * h, 3, dtype=np.int32) for _ in range(n)]
    for i in range(n):
        if keys[i]:
            orc.DecompressI(frames[i], obufs[i])
        else:
            data, sig = orc.DecompressP(frames[i], obufs[i])
            assert bool(signif[i]) == sig
            assert bool(adopted[i]) == (data is obufs[i])
        assert np.array_equal(obufs[i], to_np(dsts[i])), i
    st.close()


@pytest.mark.parametrize("bits", [16, 8])
@pytest.mark.parametrize("pieces", [2, 5])
def test_replay_rebuilds_its_block_tables_with_buffers_repeating_inside_the_group(bits, pieces):
    """A replay of a staged batch of inter frames writes its block tables again (one launch of the fused parse in its descriptor form)
    and paints from them with one temporal launch, five buffers rotating inside the group.  Same frames, flags and adoption as the
    oracle after the first decode and after two replays, with the tables poisoned before each replay (whatever comes out right was
    rebuilt).  (`pieces` only sizes the clip: round 4's parse-in-pieces option, measured slower, is gone.)"""
    if PARSE_MODE != "gpu":
        pytest.skip("on-GPU parse only")
    w, h, n = 132, 76, 97 + 32 * pieces
    frames, keys, pal = sg.msv1_clip(11 + pieces, w, h, n, bits=bits, p_mix=sg.msv1_p_mix(0.7, 25.0), key_every=n + 1)
    gpu = make_gpu(bits, w, h, pal)
    gpu.Preinit(36)
    gpu.set_option("msv1_scrub_tables", "1")
    nbuf = 5                                                   # (buffers repeat inside the group)
    dsts = [dev_buf(w * h, 3) for _ in range(nbuf)]
    st = gpu.stage_batch(frames, [dsts[i % nbuf] for i in range(n)], is_key=keys)
    assert "msv1_blocks_temporal_kernel" in st.kernels()
    orc = OracleMSVideo1(bits, w, h, pal)
    orc.Preinit(36)
    obufs = [np.full(w * h, 3, dtype=np.int32) for _ in range(nbuf)]
    want_sig, want_adopted = [], []
    for i in range(n):
        if keys[i]:
            orc.DecompressI(frames[i], obufs[i % nbuf])
            want_sig.append(None)
            want_adopted.append(None)
        else:
            data, sig = orc.DecompressP(frames[i], obufs[i % nbuf])
            want_sig.append(sig)
            want_adopted.append(data is obufs[i % nbuf])
    for run in range(3):
        for d in dsts:
            d.fill_(3)
        st.decode()
        gpu.sync()
        status, adopted, signif = st.results()
        assert status == [0] * n
        for i in range(n):
            if want_sig[i] is not None:
                assert bool(signif[i]) == want_sig[i], (run, i)
                assert bool(adopted[i]) == want_adopted[i], (run, i)
        for k in range(nbuf):
            assert np.array_equal(obufs[k], to_np(dsts[k])), (run, k)
    assert gpu.counter("lookback_fallbacks") == 0
    st.close()


@pytest.mark.parametrize("compact", ["on", "off"])
@pytest.mark.parametrize("ahead", ["on", "off"])
def test_replays_back_to_back_with_the_next_parse_running_beside_them(ahead, compact):
    """Option msv1_parse_ahead: a replay of an inter-frame batch queues the NEXT replay's table-writing parse on the codec's second stream,
    into the other of two table sets, beside its own temporal launch.  Replays queued back to back with no wait between them, tables poisoned
    before every parse, a look-back fault injected on a later replay (the parse running ahead shares the fault word: the batch is then redone
    through the three-kernel parse and must still come out right), and the batch restaged afterwards: frames equal the oracle's every time,
    on and off alike, with the replays' tables in their compact form (msv1_compact_tables) and in the 4-byte one."""
    if PARSE_MODE != "gpu":
        pytest.skip("on-GPU parse only")
    w, h, n = 320, 240, 70
    frames, keys, pal = sg.msv1_clip(23, w, h, n, p_mix=sg.msv1_p_mix(0.7, 25.0), key_every=n + 1)
    gpu = make_gpu(16, w, h, pal)
    gpu.Preinit(36)
    gpu.set_option("msv1_parse_ahead", ahead)
    gpu.set_option("msv1_compact_tables", compact)             # the replays' tables: 2 bytes per block + a base per 256 blocks, or the 4-byte ones
    gpu.set_option("msv1_scrub_tables", "1")
    nbuf = 4
    dsts = [dev_buf(w * h, 3) for _ in range(nbuf)]
    orc = OracleMSVideo1(16, w, h, pal)
    orc.Preinit(36)
    obufs = [np.full(w * h, 3, dtype=np.int32) for _ in range(nbuf)]
    for i in range(n):
        (orc.DecompressI if keys[i] else orc.DecompressP)(frames[i], obufs[i % nbuf])
    st = gpu.stage_batch(frames, [dsts[i % nbuf] for i in range(n)], is_key=keys)
    assert "msv1_blocks_temporal_kernel" in st.kernels()

    def check(tag):
        gpu.sync()
        assert st.results()[0] == [0] * n, tag
        for k in range(nbuf):
            assert np.array_equal(obufs[k], to_np(dsts[k])), (tag, k)

    st.decode()
    check("first decode")
    for burst in (1, 4, 7):                                    # replays with nothing between them: only the events order parse and reconstruction
        for d in dsts:
            d.fill_(3)
        for _ in range(burst):
            st.decode()
        check(f"burst of {burst}")
    assert gpu.counter("lookback_fallbacks") == 0
    gpu.set_option("msv1_inject_fault", "1")                   # takes effect at the next staging: restage, replay, fault at the check
    st = gpu.stage_batch(frames, [dsts[i % nbuf] for i in range(n)], is_key=keys, reuse=st)   # (jsp_restage_batch: the parse that ran ahead of the last replay is waited for first)
    for d in dsts:
        d.fill_(3)
    st.decode()
    st.decode()
    check("fault on a replay")
    assert gpu.counter("lookback_fallbacks") == 1 and "look-back fallback" in st.kernels()
    gpu.set_option("msv1_inject_fault", "0")
    for d in dsts:
        d.fill_(3)
    st.decode()
    st.decode()
    check("after the fall-back")
    st.close()


@pytest.mark.parametrize("bits", [16, 8])
def test_look_back_fault_falls_back_to_the_descriptor_kernels(bits):
    """A tile that gives up waiting for its predecessor (GPU shared with other work, profiler serialisation) is a matter of
    timing, not of the stream: the batch must then come out right through the three-kernel parse + block kernels, not as an
    error.  (The give-up itself cannot be provoked on demand: option msv1_inject_fault makes the batch's next check behave
    as if it had happened.)"""
    if PARSE_MODE != "gpu":
        pytest.skip("on-GPU parse only")
    w, h, n = 320, 240, 12
    frames, keys, pal = sg.msv1_clip(7, w, h, n, bits=bits, p_mix=sg.msv1_p_mix(0.6, 30.0), key_every=5)
    gpu = make_gpu(bits, w, h, pal)
    gpu.Preinit(36)
    gpu.set_option("msv1_inject_fault", "1")
    dsts = [dev_buf(w * h, 3) for _ in range(n)]
    st = gpu.stage_batch(frames, dsts, is_key=keys)
    assert "fallback" not in st.kernels()
    st.decode()
    gpu.sync()
    status, adopted, signif = st.results()
    assert "look-back fallback" in st.kernels()
    assert gpu.counter("lookback_fallbacks") == 1
    assert status == [0] * n
    orc = OracleMSVideo1(bits, w, h, pal)
    orc.Preinit(36)
    obufs = [np.full(w * h, 3, dtype=np.int32) for _ in range(n)]
    for i in range(n):
        if keys[i]:
            orc.DecompressI(frames[i], obufs[i])
        else:
            data, sig = orc.DecompressP(frames[i], obufs[i])
            assert bool(signif[i]) == sig
            assert bool(adopted[i]) == (data is obufs[i])
        assert np.array_equal(obufs[i], to_np(dsts[i])), i
    st.close()


@pytest.mark.parametrize("bits", [16, 8])
def test_fuzz_many_random_streams(bits):
    """A few thousand short random / mutated streams at random small sizes: oracle and HIP path must
    agree on every pixel, every significant_changes flag and every buffer identity."""
    rng = np.random.default_rng(1234 + bits)
    total = 0
    for case in range(12):
        w, h = int(rng.integers(1, 17)) * 4 + int(rng.integers(0, 4)), int(rng.integers(1, 13)) * 4 + int(rng.integers(0, 4))
        frames, keys, pal = sg.msv1_clip(3000 + case, w, h, 4, bits=bits, p_mix=sg.msv1_p_mix(0.5, 6.0))
        srcs, ks = [frames[0]], [True]
        for k in range(120):
            kind = int(rng.integers(0, 4))
            base = bytearray(frames[1 + k % 3])
            if kind == 0:      # as generated
                pass
            elif kind == 1:    # random byte flips
                for _ in range(int(rng.integers(1, 6))):
                    if base:
                        base[int(rng.integers(0, len(base)))] = int(rng.integers(0, 256))
            elif kind == 2:    # truncated
                base = base[: int(rng.integers(0, len(base) + 1))]
            else:              # pure noise
                base = bytearray(rng.integers(0, 256, size=int(rng.integers(0, 3 * len(base) + 8)), dtype=np.uint8).tobytes())
            srcs.append(bytes(base))
            ks.append(bool(rng.integers(0, 8) == 0))
        total += len(srcs)
        drive_pair(bits, w, h, srcs, ks, pal, lines=int(rng.integers(0, 12)))
    assert total > 1400


@pytest.mark.parametrize("size", [(3, 3), (2, 9), (7, 4), (3840, 2160)], ids=lambda s: f"{s[0]}x{s[1]}")
def test_degenerate_and_4k_sizes(size):
    w, h = size
    frames, keys, pal = sg.msv1_clip(90, w, h, 3, p_mix=sg.msv1_p_mix(0.5, 30.0))
    drive_pair(16, w, h, frames, keys, pal)


@pytest.mark.parametrize("bits", [16, 8])
def test_frame_buffers_that_are_not_16_byte_aligned(bits):
    """Caller buffers only have to be int32 arrays: a view 4 bytes into an allocation takes the scalar-store
    instantiations of the block kernel (and one launch per frame for inter frames)."""
    for (w, h) in [(64, 48), (320, 240)]:
        frames, keys, pal = sg.msv1_clip(7300 + bits, w, h, 8, bits=bits, p_mix=sg.msv1_p_mix(0.6, 8.0))
        drive_pair(bits, w, h, frames, keys, pal, misalign=True)
        # and as one staged batch
        gpu = make_gpu(bits, w, h, pal)
        gpu.Preinit(36)
        dsts = [dev_buf(w * h, -1, misalign=True) for _ in frames]
        st = gpu.stage_batch(frames, dsts, is_key=keys)
        st.decode()
        gpu.sync()
        orc = OracleMSVideo1(bits, w, h, pal)
        orc.Preinit(36)
        obufs = [np.full(w * h, -1, dtype=np.int32) for _ in frames]
        for i, (f, k) in enumerate(zip(frames, keys)):
            if k:
                orc.DecompressI(f, obufs[i])
            else:
                orc.DecompressP(f, obufs[i])
        _, adopted, _ = st.results()
        for i in range(len(frames)):
            if adopted[i]:
                assert np.array_equal(to_np(dsts[i]), obufs[i]), (bits, w, h, i)
        st.close()

