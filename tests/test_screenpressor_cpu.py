"""CPU tests for the ScreenPressor path: encoder -> oracle round trips (lossless identity on known
images), and the product's HOST stage (descriptor tables, re-expanded in numpy the way the HIP
kernels expand them) against the oracle."""
import numpy as np
import pytest

from jsplayer_amd import streamgen as sg
from oracle_binding import OracleAbort, OracleScreenPressor
import hoststage_binding as hs


def oracle_decode_clip(w, h, bpp, chunks, keys, lines=36):
    o = OracleScreenPressor(w, h, bpp)
    o.Preinit(lines)
    bufs = [np.zeros(w * h, np.int32) for _ in range(3)]
    frames, sigs = [], []
    for c, k in zip(chunks, keys):
        dst = next(b for b in bufs if b is not o.PreviousFrame())
        if k:
            assert o.DecompressI(c, dst) == 0
            sigs.append(None)
        else:
            data, sig = o.DecompressP(c, dst)
            sigs.append(sig)
        frames.append(o.PreviousFrame().view(np.uint32).copy())
    return frames, sigs


@pytest.mark.parametrize("version", [2, 3, 4])
@pytest.mark.parametrize("size,bpp", [((64, 48), 24), ((320, 240), 24), ((100, 52), 24), ((37, 23), 24), ((64, 48), 16)])
def test_encoder_oracle_round_trip(version, size, bpp):
    w, h = size
    chunks, keys, frames = sg.sp_clip(300 + version, w, h, 8, bpp=bpp, version=version, key_every=5,
                                      flat_at=(3,), unchanged_at=(2,))
    got, sigs = oracle_decode_clip(w, h, bpp, chunks, keys)
    for i, (g, f) in enumerate(zip(got, frames)):
        assert np.array_equal(g, f), f"frame {i} (key={keys[i]}) differs"
    assert chunks[2] == b"\x00" and sigs[2] is False


@pytest.mark.parametrize("version", [2, 4])
def test_round_trip_1080p_iframe_and_pframe(version):
    w, h = 1920, 1080
    chunks, keys, frames = sg.sp_clip(3, w, h, 2, version=version)
    got, _ = oracle_decode_clip(w, h, 24, chunks, keys)
    assert np.array_equal(got[0], frames[0]) and np.array_equal(got[1], frames[1])


def test_is_key_frame_and_headers():
    o = OracleScreenPressor(16, 16, 24)
    for b, exp in [(0x12, True), (0x11, True), (0x22, True), (0x21, True), (0x32, True), (0x31, True),
                   (0x00, False), (0x01, False), (0x42, False), (0x13, False)]:
        assert o.IsKeyFrame(bytes([b, 0, 0])) == exp
    assert o.IsKeyFrame(b"") is False
    # unknown frame type -> error_occured ; unknown version -> error_occured
    dst = np.zeros(256, np.int32)
    assert o.DecompressI(bytes([0x13, 0, 0, 0]), dst) == 2
    assert o.DecompressI(bytes([0x52, 0, 0, 0, 0, 0, 0]), dst) == 2
    # flat key frame before any coded key frame: the reference dereferences a null coder
    assert o.DecompressI(bytes([0x11, 1, 2, 3]), dst) == 3


def drive_host_stage(w, h, bpp, chunks, keys, frames, lines=36, band_rows=0):
    """Host stage -> descriptors -> numpy kernel emulation, against the oracle frame by frame."""
    host = hs.HostStage(w, h, bpp)
    host.preinit(lines)
    if band_rows:
        host.set_band_rows(band_rows)
    got, sigs = oracle_decode_clip(w, h, bpp, chunks, keys, lines)
    prev = None
    for i, (c, k) in enumerate(zip(chunks, keys)):
        d = host.decode(k, c)
        assert d["status"] == 0, i
        if d["kind"] == hs.KIND_NONE:
            assert not d["adopted"]
            cur = prev
        elif d["kind"] in (hs.KIND_FLAT, hs.KIND_INTRA):
            cur = hs.expand_iframe(d, w, h)
        else:
            cur = hs.expand_pframe(d, prev, w, h)
            assert d["significant"] == sigs[i]
        assert np.array_equal(cur, got[i]), f"frame {i} kind {d['kind']}: {(cur != got[i]).sum()} pixels differ"
        assert np.array_equal(cur, frames[i])
        prev = cur
    host.close()


@pytest.mark.parametrize("version", [2, 3, 4])
@pytest.mark.parametrize("size,bpp", [((64, 48), 24), ((320, 240), 24), ((100, 52), 24), ((37, 23), 24), ((64, 48), 16)])
def test_host_stage_descriptors_match_oracle(version, size, bpp):
    w, h = size
    chunks, keys, frames = sg.sp_clip(500 + version, w, h, 10, bpp=bpp, version=version, key_every=6,
                                      flat_at=(4,), unchanged_at=(2,))
    drive_host_stage(w, h, bpp, chunks, keys, frames, lines=4)


@pytest.mark.parametrize("band_rows", [1, 2, 5, 8, 47, 48, 1000])
def test_key_frame_bands_expand_from_their_seed_rows(band_rows):
    """Key frames cut into bands: every band rebuilt from its seed row alone gives the oracle's picture;
    the seed words are the rows above the bands (and the pixel column 0's above-left predictor wraps to)."""
    w, h = 100, 48
    chunks, keys, frames = sg.sp_clip(640, w, h, 4, version=4, key_every=2)
    drive_host_stage(w, h, 24, chunks, keys, frames, lines=4, band_rows=band_rows)
    host = hs.HostStage(w, h, 24)
    host.preinit(4)
    host.set_band_rows(band_rows)
    d = host.decode(True, chunks[0])
    nb = (h + band_rows - 1) // band_rows if band_rows < h else 1
    assert d["seeds"].size == (nb - 1) * (w + 1)
    pic = frames[0].reshape(h, w)
    for b in range(1, nb):
        sd = d["seeds"][(b - 1) * (w + 1):b * (w + 1)]
        y0 = b * band_rows
        assert np.array_equal(sd[1:], pic[y0 - 1])
        assert sd[0] == (pic[y0 - 2, w - 1] if y0 >= 2 else 0)


@pytest.mark.parametrize("size", [(100, 48), (37, 23), (600, 40), (256, 17), (260, 9)], ids=lambda s: f"{s[0]}x{s[1]}")
@pytest.mark.parametrize("span", [128, 256])
@pytest.mark.parametrize("band_rows", [0, 1, 5, 24])
def test_key_frame_tiles_stand_alone(size, band_rows, span):
    """Tile layout of key frames (one wave per band x 256-column span on the GPU): every tile is rebuilt from its
    own 4-byte records (column inside the span | value, sorted by kind, two counts per row), the seed row above its band and its
    column of left pixels only, and the mosaic is the oracle's frame.  Widths below, at, just above and well above one span (128
    or 256 columns: a record's column has 8 bits)."""
    w, h = size
    chunks, keys, frames = sg.sp_clip(660 + w, w, h, 3, version=4, key_every=1, rects=30, gradients=8)
    host = hs.HostStage(w, h, 24)
    host.preinit(4)
    host.set_iframe_layout(band_rows, span)
    for c, img in zip(chunks, frames):
        d = host.decode(True, c)
        assert d["status"] == 0 and d["kind"] == hs.KIND_INTRA
        assert np.array_equal(hs.expand_iframe_tiles(d, w, h), img)


@pytest.mark.parametrize("version", [2, 4])
def test_motion_rectangles_as_literals(version):
    """literalise_motion (what lets inter frames share a launch): afterwards no block is motion-compensated,
    the payload grew by the moved pixels (each rectangle on a 16-byte boundary), and the block table still expands to the oracle's frame."""
    w, h = 100, 52
    chunks, keys, frames = sg.sp_clip(650 + version, w, h, 6, version=version,
                                      p_mix_at={2: dict(unchanged=0.3, motion=0.6), 4: dict(unchanged=0.5, motion=0.3)})
    host = hs.HostStage(w, h, 24)
    host.preinit(4)
    prev, moved_any = None, 0
    for i, (c, k) in enumerate(zip(chunks, keys)):
        d = host.decode(k, c)
        if d["kind"] == hs.KIND_INTRA:
            cur = hs.expand_iframe(d, w, h)
        elif d["kind"] == hs.KIND_NONE:      # nothing changed in this small frame
            cur = prev
        else:
            moved = sum((int(b[3]) - int(b[1])) * (int(b[4]) - int(b[2])) for b in d["blocks"] if b[0] & hs.PB_MOTION)
            lit = host.literalise_motion(d)
            assert not any(b[0] & hs.PB_MOTION for b in lit["blocks"])
            # (every rectangle's literals start on a 16-byte boundary: up to three words of padding in front of each)
            nmoved = sum(1 for b in d["blocks"] if b[0] & hs.PB_MOTION)
            assert d["payload"].size + moved <= lit["payload"].size <= d["payload"].size + moved + 3 * nmoved
            for b in lit["blocks"]:
                if b[0] & hs.PB_DATA:
                    assert int(np.frombuffer(bytes(b[12:16]), dtype="<u4")[0]) % 4 == 0
            cur = hs.expand_pframe(lit, prev, w, h)
            assert np.array_equal(cur, hs.expand_pframe(d, prev, w, h))
            moved_any += moved
        assert np.array_equal(cur, frames[i]), i
        prev = cur
    assert moved_any > 1000


def test_generator_refuses_what_the_version_3_model_cannot_code():
    """Found by tools/fuzz_gpu.py: with f0 = 64 (version 3) the reference's table built from a 64-symbol list
    (Cx6.createFrom2) hands out more than 4096 slots once a colour context saw 60+ distinct values before its first
    repeat; symbols pushed past slot 4095 cannot be coded by any encoder, so the generator must refuse such (noise)
    content instead of emitting a stream no decoder can follow.  Version 4 (f0 = 32) codes the same images."""
    kw = dict(p_mix_at={1: dict(unchanged=0.01, motion=0.9), 2: dict(unchanged=0.02, motion=0.3)}, noise=0.9)
    with pytest.raises(RuntimeError, match="12-bit code space"):
        sg.sp_clip(916177101, 2288, 192, 3, version=3, **kw)
    chunks, keys, frames = sg.sp_clip(916177101, 2288, 192, 3, version=4, **kw)
    got, _ = oracle_decode_clip(2288, 192, 24, chunks, keys, 36)
    assert all(np.array_equal(g, f) for g, f in zip(got, frames))


def test_host_stage_rejects_what_the_reference_cannot_survive():
    host = hs.HostStage(16, 16, 24)
    assert host.decode(True, bytes([0x13, 0, 0, 0]))["status"] == 2
    assert host.decode(True, bytes([0x52, 0, 0, 0, 0, 0, 0]))["status"] == 2
    assert host.decode(True, bytes([0x11, 1, 2, 3]))["status"] == 2   # flat first: reference raises
    assert host.decode(False, bytes([1, 2, 3]))["kind"] == hs.KIND_NONE  # P before any I: ignored


@pytest.mark.parametrize("version", [2, 4])
def test_truncated_streams_agree(version):
    """A stream cut short: the v2 coder is poisoned by the missing bytes (NaN in the reference) and
    the reference never returns; v3/v4 read zeros.  Oracle and host stage must take the same exit."""
    w, h = 64, 48
    chunks, keys, frames = sg.sp_clip(700, w, h, 2, version=version)
    for cut in (1, 2, 5, 6, 9, len(chunks[0]) // 2, len(chunks[0]) - 3):
        src = chunks[0][:cut]
        o = OracleScreenPressor(w, h, 24)
        dst = np.zeros(w * h, np.int32)
        rc = o.DecompressI(src, dst)
        host = hs.HostStage(w, h, 24)
        d = host.decode(True, src)
        assert (rc == 0) == (d["status"] == 0), (version, cut, rc, d["status"])
        if rc == 0:
            assert np.array_equal(hs.expand_iframe(d, w, h), dst.view(np.uint32))


def _same_frame(a, b, tag):
    for k in ("status", "kind", "adopted", "significant", "prev_cleared", "flat_colour", "prev_pixels", "data_pixels", "stream_bytes"):
        assert a[k] == b[k], (tag, k, a[k], b[k])
    for k in ("runs", "rows", "blocks", "payload", "seeds", "tile_idx", "left"):
        assert np.array_equal(a[k], b[k]), (tag, k)


@pytest.mark.parametrize("version", [2, 4])
@pytest.mark.parametrize("threads", [2, 5])
def test_groups_of_pictures_side_by_side_equal_one_by_one(version, threads):
    """decode_frames(): the frames between one coded key frame and the next are decoded by a decoder of their own on a host
    thread of their own; every table of every frame, and the state the stream's decoder is left in, must be what decoding
    the frames one after the other gives — with flat key frames and unchanged frames inside the groups, inter frames in
    front of the first key frame, a key frame that does not decode (its group and the rest fall back to decoding in order)
    and a second batch continuing the first."""
    w, h = 320, 240
    chunks, keys, _ = sg.sp_clip(61, w, h, 26, version=version, key_every=4, flat_at=(9, 10), unchanged_at=(2, 14))
    chunks, keys = list(chunks), list(keys)
    tail, tail_keys = chunks[20:], keys[20:]
    cases = {
        "plain": (chunks[:20], keys[:20]),
        "starts with inter frames": (chunks[1:20], keys[1:20]),
        "broken key frame": (chunks[:12] + [chunks[12][:len(chunks[12]) // 3]] + chunks[13:20], keys[:20]),
    }
    for name, (fr, ks) in cases.items():
        assert sum(ks) >= 4
        seq, par = hs.HostStage(w, h, 24), hs.HostStage(w, h, 24)
        for d in (seq, par):
            d.preinit(36)
            d.set_iframe_layout(24, 256)
        one = [seq.decode(k, f) for f, k in zip(fr, ks)]
        side = par.decode_batch(fr, ks, threads)
        for i, (a, b) in enumerate(zip(one, side)):
            _same_frame(a, b, (name, i))
        # the stream goes on: the decoder the batch left behind continues like the one that saw every frame
        one2 = [seq.decode(k, f) for f, k in zip(tail, tail_keys)]
        side2 = par.decode_batch(tail, tail_keys, threads)
        for i, (a, b) in enumerate(zip(one2, side2)):
            _same_frame(a, b, (name, "continued", i))
        seq.close()
        par.close()


def test_side_by_side_literalises_what_one_by_one_does():
    w, h = 320, 240
    chunks, keys, _ = sg.sp_clip(62, w, h, 12, version=4, key_every=3, p_mix_at={i: dict(unchanged=0.5, motion=0.1) for i in range(12)})
    seq, par = hs.HostStage(w, h, 24), hs.HostStage(w, h, 24)
    side = par.decode_batch(chunks, keys, 3, literalise=True)
    for i, (f, k) in enumerate(zip(chunks, keys)):
        a = seq.decode(k, f)
        if a["kind"] == hs.KIND_INTER and side[i]["literalised"]:
            a = seq.literalise_motion(a)
        _same_frame(a, side[i], i)
    seq.close()
    par.close()


def column0_clip(version=4):
    """Three frames whose last one codes pixel (0, y0) of a data rectangle as "the pixel to the left": at x = 0 that is the last
    pixel of the row above (ScreenPressor.hx:436-444, linear index di - 1), which belongs to a block the frame has not decoded
    yet — so the decoder reads whatever its DESTINATION buffer held there.  The encoder is told (set_stale) that this is the
    picture two frames back, which is what a caller rotating two buffers hands it; frame 1 changes exactly that pixel, so one
    and two frames back differ there."""
    w, h, y0 = 64, 48, 19
    rng = sg.SplitMix64(sg.SEED_BASE + 901)
    enc = sg.SpEncoder(w, h, 24, version)
    f0 = sg.desktop_frame(rng, w, h, 24).astype(np.uint32).reshape(h, w)
    v0 = int(f0[y0 - 1, w - 1])
    f1 = f0.copy()
    f1[y0 - 1, w - 1] = (v0 ^ 0x00F0F0F0) & 0xFFFFFF
    f2 = f1.copy()
    f2[y0, 0] = v0
    assert v0 != int(f1[y0, 0]) and v0 != int(f2[y0 - 1, 0])
    chunks = [enc.encode_i(f0), enc.encode_p(f1)]
    enc.set_stale(f0)
    chunks.append(enc.encode_p(f2))
    enc.close()
    return w, h, y0, chunks, [f0, f1, f2]


def oracle_with_rotation(w, h, chunks, nbuf, fill=0x123456):
    o = OracleScreenPressor(w, h, 24)
    o.Preinit(36)
    bufs = [np.full(w * h, fill, np.int32) for _ in range(nbuf)]
    o.DecompressI(chunks[0], bufs[0])
    o.DecompressP(chunks[1], bufs[1])
    o.DecompressP(chunks[2], bufs[2 % nbuf])
    return o.PreviousFrame().view(np.uint32).reshape(h, w).copy()


@pytest.mark.parametrize("version", [2, 4])
def test_left_of_column_zero_is_the_destinations_old_content(version):
    """The one place where an inter frame READS its destination: the reference reads the caller's buffer
    there, so its result depends on how the caller rotates its buffers.  With two buffers in rotation it sees the picture two
    frames back (and the stream decodes to what the encoder meant); with three it reads older content and differs in exactly
    that pixel.  The product's host stage is told what the destination holds in its last column (the codec keeps track,
    SpCodec::before / after) and then decodes what the reference decodes, whatever the rotation; told nothing, it reads its
    own shadow of the position — the two-buffer picture."""
    w, h, y0, chunks, frames = column0_clip(version)
    two = oracle_with_rotation(w, h, chunks, 2)
    assert np.array_equal(two, frames[2])
    three = oracle_with_rotation(w, h, chunks, 3)
    ys, xs = np.nonzero(three != two)
    assert (ys.tolist(), xs.tolist()) == ([y0], [0]) and int(three[y0, 0]) == 0x123456   # the third buffer's fill shows through
    # the product's host stage -> descriptors -> numpy kernel emulation: as the reference with two buffers
    host = hs.HostStage(w, h, 24)
    host.preinit(36)
    prev = None
    for i, c in enumerate(chunks):
        d = host.decode(i == 0, c)
        assert d["status"] == 0
        prev = hs.expand_iframe(d, w, h) if i == 0 else hs.expand_pframe(d, prev, w, h)
    host.close()
    assert np.array_equal(np.asarray(prev).view(np.uint32).reshape(h, w), two)
    # ... and told what a third buffer holds (its fill): as the reference with three buffers
    host = hs.HostStage(w, h, 24)
    host.preinit(36)
    prev = None
    for i, c in enumerate(chunks):
        host.set_dst_column(np.full(h, 0x123456, np.int32) if i == 2 else None)
        d = host.decode(i == 0, c)
        assert d["status"] == 0
        prev = hs.expand_iframe(d, w, h) if i == 0 else hs.expand_pframe(d, prev, w, h)
    host.close()
    assert np.array_equal(np.asarray(prev).view(np.uint32).reshape(h, w), three)


