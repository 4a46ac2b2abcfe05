"""BASELINE.json configs[0]: an MSVideo1 320x240 clip written to a real RIFF/AVI file, parsed back and
pushed through the IVideoCodec plumbing (Manager-equivalent loop).  CPU leg: the oracle codecs
("Haxe CPU reference path, no GPU").  GPU leg: the HIP codecs on the same bytes, frame for frame."""
import numpy as np
import pytest

from jsplayer_amd import avi, player
from jsplayer_amd import streamgen as sg
from oracle_binding import OracleMSVideo1, OracleScreenPressor


class _Res:
    def __init__(self, data, sig):
        self.data_pnt, self.significant_changes = data, sig


class _OrcAdapter:
    """Gives the oracle classes the exact IVideoCodec return shapes."""

    def __init__(self, o):
        self.o = o

    def __getattr__(self, k):
        return getattr(self.o, k)

    def DecompressP(self, src, dst):
        return _Res(*self.o.DecompressP(src, dst))


ORACLE_CLASSES = (lambda w, h: _OrcAdapter(OracleMSVideo1(16, w, h)),
                  lambda w, h, pal: _OrcAdapter(OracleMSVideo1(8, w, h, pal)),
                  lambda w, h, bpp: _OrcAdapter(OracleScreenPressor(w, h, bpp)))


def config0_clip(bits=16, n=100):
    # frame 0 all coded, later frames ~70 % skipped (geometric runs, mean 40), mix M1
    frames, keys, pal = sg.msv1_clip(1, 320, 240, n, bits=bits, p_mix=sg.msv1_p_mix(0.70, 40.0))
    return frames, pal


@pytest.mark.parametrize("bits", [16, 8])
def test_avi_round_trip_and_cpu_reference_path(bits):
    frames, pal = config0_clip(bits, 100)
    odd = [f + (b"\x07" if i % 7 == 3 else b"") for i, f in enumerate(frames)]   # some odd-sized chunks
    blob = avi.write_avi(320, 240, odd, fourcc=b"CRAM", bpp=bits, fps=15.0, palette=pal)
    vi, got = avi.read_avi(blob)
    assert (vi.X, vi.Y, vi.bpp, vi.nframes) == (320, 240, bits, 100)
    assert vi.codec == (avi.CODEC_MSVC16 if bits == 16 else avi.CODEC_MSVC8)
    assert abs(vi.fps - 15.0) < 0.01
    if bits == 8:
        assert vi.palette[:len(pal)] == pal
    # the codec sees the chunk padded to even length (ParserUtils.hx:24-27)
    assert [len(g) for g in got] == [(len(f) + 1) & ~1 for f in odd]
    assert all(g[:len(f)] == f for g, f in zip(got, odd))
    # decode through the Manager-equivalent loop on the oracle ("CPU reference path")
    dec = player.make_decoder(vi, ORACLE_CLASSES)
    mgr = player.Manager(vi, dec, lambda n: np.zeros(n, dtype=np.int32))
    log = mgr.play(got)
    assert len(log) == 100 and log[0].key and log[0].significant_changes is True
    assert not any(d.key for d in log[1:])
    # direct decode of the raw frames gives the same pictures (the pad byte changes nothing here)
    ref = OracleMSVideo1(bits, 320, 240, pal)
    ref.Preinit(36)
    bufs = [np.zeros(320 * 240, np.int32) for _ in range(2)]
    for i, f in enumerate(frames):
        dst = bufs[0] if ref.PreviousFrame() is bufs[1] else bufs[1]
        (ref.DecompressI if i == 0 else ref.DecompressP)(f, dst)
    assert np.array_equal(ref.PreviousFrame(), mgr.buffers[log[-1].buffer_index])


def test_screenpressor_avi_defaults_to_screenpressor_codec():
    chunks, keys, frames = sg.sp_clip(5, 64, 48, 5, version=4, unchanged_at=(2,))
    blob = avi.write_avi(64, 48, chunks, fourcc=b"SCPR", bpp=24)
    vi, got = avi.read_avi(blob)
    assert vi.codec == avi.CODEC_SCREENPRESSOR and vi.bpp == 24
    dec = player.make_decoder(vi, ORACLE_CLASSES)
    mgr = player.Manager(vi, dec, lambda n: np.zeros(n, dtype=np.int32))
    log = mgr.play(got)
    for d, f in zip(log, frames):
        pass
    assert np.array_equal(mgr.buffers[log[-1].buffer_index].view(np.uint32), frames[-1])
    # the "no changes" frame keeps showing the previous slot
    assert log[2].buffer_index == log[1].buffer_index and log[2].significant_changes is False


@pytest.mark.gpu
@pytest.mark.parametrize("bits", [16, 8])
def test_avi_clip_gpu_matches_cpu_reference_path(bits):
    import torch
    from jsplayer_amd import MSVideo1_16bit, MSVideo1_8bit, ScreenPressor
    frames, pal = config0_clip(bits, 100)
    blob = avi.write_avi(320, 240, frames, fourcc=b"CRAM", bpp=bits, palette=pal)
    vi, got = avi.read_avi(blob)
    cpu = player.Manager(vi, player.make_decoder(vi, ORACLE_CLASSES), lambda n: np.zeros(n, dtype=np.int32))
    gpu = player.Manager(vi, player.make_decoder(vi, (MSVideo1_16bit, MSVideo1_8bit, ScreenPressor)),
                         lambda n: torch.zeros(n, dtype=torch.int32, device="cuda"))
    for i, f in enumerate(got):
        a = cpu.worker(f, i, None)
        b = gpu.worker(f, i, None)
        assert (a.key, a.buffer_index, a.significant_changes, a.state) == (b.key, b.buffer_index, b.significant_changes, b.state)
        assert np.array_equal(cpu.buffers[a.buffer_index], gpu.buffers[b.buffer_index].cpu().numpy()), i


@pytest.mark.gpu
@pytest.mark.parametrize("what", ["msvc16", "msvc8", "screenpressor"])
@pytest.mark.parametrize("depth,prefetch", [(1, 0), (3, 0), (3, 20000)], ids=["1", "3", "3-prefetched"])
def test_python_player_pipelined_shows_the_same_pictures(what, depth, prefetch):
    """Manager.play_pipelined (DecompressI_async / DecompressP_async / wait, `depth` frames in flight) writes the log
    Manager.play writes through the oracle — key flags, significance, states — and every frame shows the same picture
    (the pool is larger, so slot numbers may differ); also with the frames taken to the device in ranges ahead of them."""
    import torch
    from jsplayer_amd import MSVideo1_16bit, MSVideo1_8bit, ScreenPressor
    if what == "screenpressor":
        chunks, keys, _ = sg.sp_clip(32, 320, 240, 14, version=4, key_every=5, unchanged_at=(2, 7))
        blob = avi.write_avi(320, 240, chunks, fourcc=b"SCPR", bpp=24, key_flags=keys)
    else:
        bits = 16 if what == "msvc16" else 8
        frames, pal = config0_clip(bits, 40)
        blob = avi.write_avi(320, 240, frames, fourcc=b"CRAM", bpp=bits, palette=pal)
    vi, got = avi.read_avi(blob)
    shown_cpu, shown_gpu = [], []
    cpu = player.Manager(vi, player.make_decoder(vi, ORACLE_CLASSES), lambda n: np.zeros(n, dtype=np.int32))
    cpu.play(got, on_frame=lambda d, buf: shown_cpu.append(buf.copy()))
    dec = player.make_decoder(vi, (MSVideo1_16bit, MSVideo1_8bit, ScreenPressor))
    if what != "screenpressor":
        dec.set_option("msv1_parse", "gpu")
    gpu = player.Manager(vi, dec, lambda n: torch.zeros(n, dtype=torch.int32, device="cuda"), num_buffers=player.NUM_BUFFERS + depth)
    # (prefetch: the frames in one pinned arena that goes to the device in ranges of ~20 KB — a few frames each — ahead of them)
    gpu.play_pipelined(got, depth=depth, on_frame=lambda d, buf: shown_gpu.append(buf.cpu().numpy()), prefetch_bytes=prefetch)
    if prefetch and what != "screenpressor":
        assert dec.counter("prefetched_frames") >= len(got) // 2
    assert len(cpu.log) == len(gpu.log) == len(got)
    for a, b, pa, pb in zip(cpu.log, gpu.log, shown_cpu, shown_gpu):
        assert (a.index, a.key, a.significant_changes, a.state) == (b.index, b.key, b.significant_changes, b.state)
        assert np.array_equal(pa, pb), a.index


@pytest.mark.gpu
def test_display_convert_and_frames_differ_kernels():
    """jsp_display_convert / jsp_frames_differ against the oracle's restatement of Manager.fill_bitmap_data
    (Manager.hx:325-390, all four conversions, with and without the row flip) and of the pixel loop of
    frames_differ_significantly (Manager.hx:413-419); pixel values cover all 32 bits (the shifts drop high bits)."""
    import torch
    from jsplayer_amd import codec as cm
    from oracle_binding import orc_display_convert, orc_frames_differ
    rng = np.random.default_rng(3)
    for (w, h) in [(1920, 1080), (321, 7), (64, 48)]:
        src = rng.integers(0, 1 << 32, size=w * h, dtype=np.uint64).astype(np.uint32).view(np.int32)
        t = torch.from_numpy(src).cuda()
        out = torch.empty_like(t)
        for mode in (cm.DISPLAY_CANVAS, cm.DISPLAY_CANVAS_RGB15, cm.DISPLAY_SETPIXELS, cm.DISPLAY_SETPIXELS_RGB15):
            for flip in (False, True):
                cm.display_convert(t, out, w, h, mode, flip)
                torch.cuda.synchronize()
                assert np.array_equal(out.cpu().numpy(), orc_display_convert(src, w, h, mode, flip)), (w, h, mode, flip)
        b = t.clone()
        for first, poke in [(0, None), (0, w * h - 1), (w * h - 1, w * h - 1), (6, 5), (5, 5), (36 * w if 36 * w < w * h else 0, w * h // 2)]:
            if poke is not None:
                b[poke] += 1
            torch.cuda.synchronize()
            assert cm.frames_differ(t, b, first, w * h) == orc_frames_differ(src, b.cpu().numpy(), first, w * h), (w, h, first, poke)
            if poke is not None:
                b[poke] -= 1


def test_oracle_display_and_differ_known_answers():
    """The oracle's own restatement against hand-worked values of the four formulas (Manager.hx:340,351,370,379)."""
    from oracle_binding import orc_display_convert, orc_frames_differ
    px = np.array([0x00112233, 0x00FFEEDD, 0x12345678, 0x0000001F], dtype=np.uint32).view(np.int32)
    u = lambda a: a.view(np.uint32).tolist()
    assert u(orc_display_convert(px, 4, 1, 0, False)) == [0xFF332211, 0xFFDDEEFF, 0xFF785634, 0xFF1F0000]
    assert u(orc_display_convert(px, 4, 1, 1, False)) == [0xFF891198, 0xFFFF76E8, 0xFFA2B3C0, 0xFF0000F8]
    assert u(orc_display_convert(px, 4, 1, 2, False)) == [0xFF112233, 0xFFFFEEDD, 0xFF345678, 0xFF00001F]
    assert u(orc_display_convert(px, 4, 1, 3, False)) == [0x89119800, 0xFF76E800, 0xA2B3C000, 0x0000F800]
    assert u(orc_display_convert(px, 2, 2, 2, True)) == [0xFF345678, 0xFF00001F, 0xFF112233, 0xFFFFEEDD]
    other = px.copy()
    other[1] ^= 1
    assert not orc_frames_differ(px, px.copy(), 0, 4) and orc_frames_differ(px, other, 0, 4) and not orc_frames_differ(px, other, 2, 4)


@pytest.mark.parametrize("per_ix", [0, 7, 1000])
def test_index_key_flags_and_random_access(per_ix):
    """idx1 (per_ix=0) and OpenDML indx/ix00: offsets, sizes (odd ones keep their pad byte), key flags and an
    empty frame come back through the index exactly as written; the sequential walk sees the same blobs."""
    rng = np.random.default_rng(5)
    frames = [rng.integers(0, 256, size=int(n), dtype=np.uint8).tobytes() for n in (10, 33, 0, 64, 7, 7, 128, 1, 90, 2, 31)]
    keys = [True, False, False, True, False, False, False, True, False, False, False]
    blob = avi.write_avi(64, 48, frames, fourcc=b"SCPR", bpp=24, key_flags=keys, opendml_frames_per_ix=per_ix)
    index = avi.read_index(blob)
    assert [e.key for e in index] == keys and [e.size for e in index] == [len(f) for f in frames]
    for e, f in zip(index, frames):
        assert blob[e.offset:e.offset + 4] == b"00dc" and blob[e.offset + 8:e.offset + 8 + len(f)] == f
    vi, got, got_keys = avi.read_avi_indexed(blob)
    assert got_keys == keys and vi.nframes == len(frames)
    assert got == [f + (b"\0" if len(f) & 1 else b"") for f in frames]
    assert avi.read_avi(blob)[1] == got                    # ix00 chunks inside movi are not frames


def test_no_index_falls_back_to_sequential_walk():
    frames = [b"\x01\x02", b"\x03\x04\x05\x06"]
    blob = avi.write_avi(16, 8, frames)
    cut = blob[:blob.index(b"idx1")]
    cut = cut[:4] + (len(cut) - 8).to_bytes(4, "little") + cut[8:]
    assert avi.read_index(cut) is None
    _, got, keys = avi.read_avi_indexed(cut)
    assert got == frames and keys is None   # the caller asks the decoder (DataLoaderAVISeq.hx:45)


def test_indexed_playback_matches_sequential_playback():
    """The same ScreenPressor clip played from the OpenDML index (key flags from the index) and from the
    sequential walk (key flags from IsKeyFrame) shows the same pictures, frame for frame."""
    chunks, keys, frames = sg.sp_clip(6, 64, 48, 9, version=3, unchanged_at=(2,), key_every=4)
    blob = avi.write_avi(64, 48, chunks, fourcc=b"SCPR", bpp=24, key_flags=keys, opendml_frames_per_ix=4)
    vi, got, idx_keys = avi.read_avi_indexed(blob)
    assert idx_keys == [bool(k) for k in keys]
    shown = []
    for flags in (idx_keys, None):
        mgr = player.Manager(vi, player.make_decoder(vi, ORACLE_CLASSES), lambda n: np.zeros(n, dtype=np.int32))
        pics = []
        mgr.play(got, on_frame=lambda d, buf: pics.append(buf.copy()), key_flags=flags)
        shown.append(pics)
    assert len(shown[0]) == 9
    for a, b, f in zip(shown[0], shown[1], frames):
        assert np.array_equal(a, b) and np.array_equal(a.view(np.uint32), f)


@pytest.mark.gpu
@pytest.mark.parametrize("what", ["msvc16", "msvc8", "screenpressor"])
def test_cpp_player_over_the_c_abi(what, tmp_path):
    """examples/jsp_play: a C++ host that links nothing but libjsplayer_amd.so (no Python, no torch) plays an AVI file
    through the C ABI with the Manager's buffer discipline; per frame it must show what the Manager-equivalent loop
    shows on the oracle codecs — same key flags, same pool slots, same significant_changes, same picture (CRC-32)."""
    import os
    import subprocess
    import zlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "jsp_play")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(root, "examples")])
    if what == "screenpressor":
        chunks, keys, _ = sg.sp_clip(31, 320, 240, 12, version=4, key_every=5, unchanged_at=(2, 7))
        blob = avi.write_avi(320, 240, chunks, fourcc=b"SCPR", bpp=24, key_flags=keys)   # the player takes the index's flags
    else:
        bits = 16 if what == "msvc16" else 8
        frames, pal = config0_clip(bits, 30)
        probe = ORACLE_CLASSES[0](320, 240) if bits == 16 else ORACLE_CLASSES[1](320, 240, pal)
        blob = avi.write_avi(320, 240, frames, fourcc=b"CRAM", bpp=bits, palette=pal,
                             key_flags=[i == 0 or probe.IsKeyFrame(f) for i, f in enumerate(frames)])
    path = tmp_path / "clip.avi"
    path.write_bytes(blob)
    res = subprocess.run([exe, str(path)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert res.returncode == 0, res.stderr.decode()
    lines = [l.split() for l in res.stdout.decode().splitlines()]
    vi, got = avi.read_avi(blob)
    mgr = player.Manager(vi, player.make_decoder(vi, ORACLE_CLASSES), lambda n: np.zeros(n, dtype=np.int32))
    shown = []
    mgr.play(got, on_frame=lambda d, buf: shown.append((d, zlib.crc32(buf.tobytes()))))
    assert len(lines) == len(shown) == len(got)
    for ln, (d, crc) in zip(lines, shown):
        assert int(ln[0]) == d.index and ln[1] == ("key" if d.key else "inter"), ln
        assert int(ln[2]) == d.buffer_index, ln
        assert int(ln[3]) == int(bool(d.significant_changes)), ln
        assert int(ln[4], 16) == crc, ln


@pytest.mark.gpu
@pytest.mark.parametrize("what", ["msvc16", "msvc8", "screenpressor"])
@pytest.mark.parametrize("batch", [1, 7, 64])
def test_cpp_player_batched_shows_the_same_pictures(what, batch, tmp_path):
    """examples/jsp_play --batch B: the file through jsp_stage_batch / jsp_staged_decode, B frames at a time (for
    ScreenPressor the host stage takes the batch's groups of pictures side by side), shows frame for frame the picture
    the synchronous loop shows."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "jsp_play")
    if what == "screenpressor":
        chunks, keys, _ = sg.sp_clip(33, 320, 240, 20, version=4, key_every=4, unchanged_at=(2, 9), flat_at=(6,))
        blob = avi.write_avi(320, 240, chunks, fourcc=b"SCPR", bpp=24, key_flags=keys)
    else:
        bits = 16 if what == "msvc16" else 8
        frames, pal = config0_clip(bits, 30)
        probe = ORACLE_CLASSES[0](320, 240) if bits == 16 else ORACLE_CLASSES[1](320, 240, pal)
        blob = avi.write_avi(320, 240, frames, fourcc=b"CRAM", bpp=bits, palette=pal,
                             key_flags=[i == 0 or probe.IsKeyFrame(f) for i, f in enumerate(frames)])
    path = tmp_path / "clip.avi"
    path.write_bytes(blob)
    outs = []
    for extra in ([], ["--batch", str(batch)]):
        res = subprocess.run([exe, str(path)] + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        assert res.returncode == 0, res.stderr.decode()
        outs.append([l.split() for l in res.stdout.decode().splitlines()])
    assert len(outs[0]) == len(outs[1]) > 0
    for a, b in zip(*outs):
        assert (a[0], a[1], a[-1]) == (b[0], b[1], b[-1]), (a, b)   # index, key / inter, CRC-32 of the picture shown


@pytest.mark.gpu
@pytest.mark.parametrize("what", ["msvc16", "msvc8", "screenpressor"])
def test_cpp_player_pipelined_shows_the_same_pictures(what, tmp_path):
    """examples/jsp_play --pipelined: the same loop over jsp_decompress_*_async / jsp_wait with three frames in flight and
    the file's bytes in pinned memory shows, frame for frame, what the synchronous loop shows (the pool is larger, so slot
    numbers may differ); its --quiet form plays several streams at once and reports a rate."""
    import json
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "jsp_play")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(root, "examples")])
    if what == "screenpressor":
        chunks, keys, _ = sg.sp_clip(31, 320, 240, 12, version=4, key_every=5, unchanged_at=(2, 7))
        blob = avi.write_avi(320, 240, chunks, fourcc=b"SCPR", bpp=24, key_flags=keys)   # the player takes the index's flags
    else:
        bits = 16 if what == "msvc16" else 8
        frames, pal = config0_clip(bits, 30)
        probe = ORACLE_CLASSES[0](320, 240) if bits == 16 else ORACLE_CLASSES[1](320, 240, pal)
        blob = avi.write_avi(320, 240, frames, fourcc=b"CRAM", bpp=bits, palette=pal,
                             key_flags=[i == 0 or probe.IsKeyFrame(f) for i, f in enumerate(frames)])
    path = tmp_path / "clip.avi"
    path.write_bytes(blob)
    outs = []
    # (--prefetch MB: the file goes to the device in ranges of that size ahead of the frames, jsp_prefetch — here ranges of a few frames)
    for extra in ([], ["--pipelined", "--depth", "3"], ["--pipelined", "--depth", "3", "--prefetch", "0.02"]):
        res = subprocess.run([exe, str(path)] + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        assert res.returncode == 0, res.stderr.decode()
        outs.append([l.split() for l in res.stdout.decode().splitlines()])
    assert len(outs[0]) == len(outs[1]) == len(outs[2]) > 0
    for a, b, c in zip(*outs):
        assert a[0] == b[0] and a[1] == b[1] and a[3] == b[3] and a[4] == b[4], (a, b)
        assert b == c, (b, c)
    res = subprocess.run([exe, str(path), "--pipelined", "--quiet", "--streams", "3", "--repeat", "2", "--depth", "4", "--prefetch", "0.05"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert res.returncode == 0, res.stderr.decode()
    rate = json.loads(res.stdout.decode())
    assert rate["streams"] == 3 and rate["frames"] == 3 * 2 * len(outs[0]) and rate["mpixels_per_s"] > 0


@pytest.mark.gpu
def test_cpp_player_shards_streams_over_devices(tmp_path):
    """examples/jsp_play --devices: independent streams sharded one per listed device inside one process (a host thread, a codec
    instance and a frame pool each; stream s plays file s on devices[s mod G]), the per-device counters summed through
    jsp_reduce_counters.  With the one device of this box listed twice every stream must print exactly what a single-device run
    of its file prints; the counters must add up; and the reduce itself must go through RCCL when a communicator can be had."""
    import ctypes as C
    import json
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "jsp_play")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(root, "examples")])
    paths = []
    chunks, keys, _ = sg.sp_clip(41, 320, 240, 10, version=4, key_every=4, unchanged_at=(2,))
    paths.append(tmp_path / "a.avi")
    paths[-1].write_bytes(avi.write_avi(320, 240, chunks, fourcc=b"SCPR", bpp=24, key_flags=keys))
    frames, pal = config0_clip(16, 24)
    probe = ORACLE_CLASSES[0](320, 240)
    paths.append(tmp_path / "b.avi")
    paths[-1].write_bytes(avi.write_avi(320, 240, frames, fourcc=b"CRAM", bpp=16, palette=pal, key_flags=[i == 0 or probe.IsKeyFrame(f) for i, f in enumerate(frames)]))
    single = []
    for p in paths:
        res = subprocess.run([exe, str(p), "--pipelined", "--depth", "3"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        assert res.returncode == 0, res.stderr.decode()
        single.append(res.stdout.decode().splitlines())
    res = subprocess.run([exe, ",".join(str(p) for p in paths), "--pipelined", "--depth", "3", "--devices", "0,0", "--streams", "3"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert res.returncode == 0, res.stderr.decode()
    out = res.stdout.decode().splitlines()
    streams, cur = {}, None
    for ln in out:
        if ln.startswith("# stream"):
            cur = int(ln.split()[2])
            assert ln.split()[3:5] == ["device", "0"]
            streams[cur] = []
        elif ln.startswith("# total"):
            total = ln.split()
        elif ln[:1].isdigit():               # (RCCL prints a version banner of its own on stdout when the communicator is made)
            streams[cur].append(ln)
    assert sorted(streams) == [0, 1, 2]
    assert streams[0] == single[0] and streams[1] == single[1] and streams[2] == single[0]      # stream 2 plays file 0 again
    nframes = 2 * len(single[0]) + len(single[1])
    assert int(total[3]) == nframes and int(total[5]) == nframes * 320 * 240 and total[7] in ("rccl", "host")
    # throughput form: the JSON line carries the per-device counters
    res = subprocess.run([exe, ",".join(str(p) for p in paths), "--pipelined", "--quiet", "--devices", "0,0", "--streams", "4", "--repeat", "2"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert res.returncode == 0, res.stderr.decode()
    line = json.loads(res.stdout.decode().strip().splitlines()[-1])
    assert line["devices"] == [0, 0] and sum(line["per_device_frames"]) == line["frames"] == 2 * (2 * len(single[0]) + 2 * len(single[1]))
    assert line["total_pixels"] == line["frames"] * 320 * 240
    assert line["async_reruns"] == 0, "a frame the GPU could not settle alone was re-run on the host: see DESIGN.md 3.3"
    # BASELINE.json configs[4] in one process: eight streams over eight listed devices (here the one device eight times): a host thread, a
    # codec instance and a pool per stream, stream s -> devices[s mod 8]; every stream shows what a single-device run of its file shows
    eight = "0,0,0,0,0,0,0,0"
    res = subprocess.run([exe, ",".join(str(p) for p in paths), "--pipelined", "--depth", "3", "--devices", eight, "--streams", "8"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=180)
    assert res.returncode == 0, res.stderr.decode()
    streams, cur, total = {}, None, None
    for ln in res.stdout.decode().splitlines():
        if ln.startswith("# stream"):
            cur = int(ln.split()[2])
            streams[cur] = []
        elif ln.startswith("# total"):
            total = ln.split()
        elif ln[:1].isdigit():
            streams[cur].append(ln)
    assert sorted(streams) == list(range(8))
    for s_ in range(8):
        assert streams[s_] == single[s_ % 2], f"stream {s_}"
    nframes8 = 4 * len(single[0]) + 4 * len(single[1])
    assert int(total[3]) == nframes8 and int(total[5]) == nframes8 * 320 * 240
    res = subprocess.run([exe, ",".join(str(p) for p in paths), "--pipelined", "--quiet", "--devices", eight, "--streams", "8", "--repeat", "3"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=180)
    assert res.returncode == 0, res.stderr.decode()
    line = json.loads(res.stdout.decode().strip().splitlines()[-1])
    assert line["devices"] == [0] * 8 and len(line["per_device_frames"]) == 8
    assert line["per_device_frames"] == [3 * len(single[s_ % 2]) for s_ in range(8)] and sum(line["per_device_frames"]) == line["frames"]
    assert line["async_reruns"] == 0
    bad = subprocess.run([exe, str(paths[0]), "--pipelined", "--devices", "0,99"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert bad.returncode == 2
    # the collective itself: one rank per distinct device; on this box one device, so a communicator of one rank — the all-reduce runs on the GPU
    from jsplayer_amd import _native as N
    lib = N.lib()
    assert lib.jsp_device_count() >= 1
    devs, per = (C.c_int * 2)(0, 0), (C.c_uint64 * 4)(5, 500, 7, 700)
    tot, via = (C.c_uint64 * 2)(), C.c_int(0)
    assert lib.jsp_reduce_counters(devs, 2, per, tot, C.byref(via)) == 0 and list(tot) == [12, 1200]
    assert via.value == 1, "RCCL all-reduce not used: " + lib.jsp_shard_last_error().decode()


@pytest.mark.gpu
@pytest.mark.parametrize("how", ["play", "pipelined", "pipelined_workers"])
def test_key_frames_compared_with_the_frame_before_them_while_they_decode(how):
    """Manager.frames_differ_significantly (Manager.hx:392-421) without a pass of the Manager's own: option "key_frame_compare"
    has the codec compare every key frame with the frame before it as it decodes it (ScreenPressor's host stage holds both
    pictures; otherwise a compare queued behind the frame's kernels).  A clip built for the pixel loop: a key frame that repeats
    the picture before it (behind an inter frame, so the byte shortcut does not apply) -> no change; a key frame that differs
    only in the rows the Manager ignores -> no change; one that differs below them -> change; flat key frames; key frames
    back to back.  The log must be the one the Manager writes over the CPU oracle with its own numpy compare."""
    import torch
    from jsplayer_amd import MSVideo1_16bit, MSVideo1_8bit, ScreenPressor
    w, h = 320, 240
    rng = sg.SplitMix64(sg.SEED_BASE + 777)
    enc = sg.SpEncoder(w, h, 24, 4)
    a = sg.desktop_frame(rng, w, h, 24).astype(np.uint32).reshape(h, w)
    top = a.copy()
    top[5:20, 10:200] ^= 0x00101010          # rows 5..19 only (the buffer is bottom-up: rows below INSIGNIFICANT_LINES = 36 do not count)
    low = a.copy()
    low[100:120, 30:90] ^= 0x00202020
    seq = [("i", a), ("p", a), ("i", a), ("p", a), ("i", top), ("p", top), ("i", low), ("i", low), ("flat", 0x336699), ("p", None), ("i", a)]
    chunks, keys = [], []
    cur = None
    for kind, img in seq:
        if kind == "i":
            chunks.append(enc.encode_i(img)); keys.append(True); cur = img
        elif kind == "flat":
            chunks.append(enc.encode_flat(img)); keys.append(True); cur = enc.current().reshape(h, w).copy()
        else:
            chunks.append(enc.encode_p(cur)); keys.append(False)
    enc.close()
    blob = avi.write_avi(w, h, chunks, fourcc=b"SCPR", bpp=24, key_flags=keys)
    vi, got = avi.read_avi(blob)
    cpu = player.Manager(vi, player.make_decoder(vi, ORACLE_CLASSES), lambda n: np.zeros(n, dtype=np.int32))
    cpu.play(got, key_flags=keys)
    depth = 3
    # (worker threads: a key frame opens a group of its own, whose worker may queue its kernels — and the compare behind them — before the
    # group in front has queued the picture it is compared with; the compare waits for that.  Several runs: it is a matter of timing.)
    for attempt in range(8 if how == "pipelined_workers" else 1):
        dec = player.make_decoder(vi, (MSVideo1_16bit, MSVideo1_8bit, ScreenPressor))
        if how != "play":
            dec.set_option("sp_async_threads", "4" if how == "pipelined_workers" else "1")
        gpu = player.Manager(vi, dec, lambda n: torch.zeros(n, dtype=torch.int32, device="cuda"), num_buffers=player.NUM_BUFFERS + depth)
        assert gpu._fused_compare
        if how == "play":
            gpu.play(got, key_flags=keys)
        else:
            gpu.play_pipelined(got, depth=depth, key_flags=keys)
        assert [(d.index, d.key, d.significant_changes, d.state) for d in cpu.log] == [(d.index, d.key, d.significant_changes, d.state) for d in gpu.log], attempt
        dec.StopAndClean()
    sig = {d.index: d.significant_changes for d in cpu.log if d.key}
    assert sig[2] is False and sig[4] is False and sig[6] is True and sig[7] is False and sig[8] is True and sig[10] is True
    # the same through MSVideo1 (16-bit): key frames behind inter frames, decoded through the asynchronous one-launch path
    frames, pal = config0_clip(16, 30)
    probe = ORACLE_CLASSES[0](320, 240)
    mk = [i == 0 or probe.IsKeyFrame(f) for i, f in enumerate(frames)]
    frames = list(frames[:6]) + [frames[0]] + list(frames[6:12]) + [frames[0], frames[0]]      # key frames again, behind inter frames
    mk = mk[:6] + [True] + mk[6:12] + [True, True]
    blob = avi.write_avi(320, 240, frames, fourcc=b"CRAM", bpp=16, palette=pal, key_flags=mk)
    vi, got = avi.read_avi(blob)
    cpu = player.Manager(vi, player.make_decoder(vi, ORACLE_CLASSES), lambda n: np.zeros(n, dtype=np.int32))
    cpu.play(got, key_flags=mk)
    dec = player.make_decoder(vi, (MSVideo1_16bit, MSVideo1_8bit, ScreenPressor))
    gpu = player.Manager(vi, dec, lambda n: torch.zeros(n, dtype=torch.int32, device="cuda"), num_buffers=player.NUM_BUFFERS + depth)
    if how == "play":
        gpu.play(got, key_flags=mk)
    else:
        gpu.play_pipelined(got, depth=depth, key_flags=mk)
    assert [(d.index, d.key, d.significant_changes, d.state) for d in cpu.log] == [(d.index, d.key, d.significant_changes, d.state) for d in gpu.log]
    assert any(d.key and d.index > 0 and d.significant_changes is True for d in cpu.log)
