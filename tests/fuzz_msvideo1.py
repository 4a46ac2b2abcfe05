#!/usr/bin/env python3
"""Randomised GPU campaign for MSVideo1 against the oracle (it lives under tests/ because it uses the oracle):
random geometry (multiples of 4 or not), depth, clip structure, skip mixes, mutated / truncated / random frames, host and
on-GPU parse, device / host / misaligned buffers, per-call API and staged batches.  Not collected by pytest.

    python tests/fuzz_msvideo1.py [seconds] [seed]
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    import test_msvideo1_gpu as T
    from jsplayer_amd import streamgen as sg
    from oracle_binding import OracleMSVideo1
    rng = np.random.default_rng(seed)
    t0, clips, nframes, bad = time.time(), 0, 0, 0
    while time.time() - t0 < budget:
        w = int(rng.choice([int(rng.integers(1, 120)) * 4, int(rng.integers(4, 500)), int(rng.integers(100, 481)) * 4]))
        h = int(rng.choice([int(rng.integers(1, 70)) * 4, int(rng.integers(4, 300)), int(rng.integers(60, 271)) * 4]))   # up to 1920x1080: dozens of 16 KiB tiles per frame
        bits = int(rng.choice([16, 8]))
        n = int(rng.integers(2, 12))
        p_mix = sg.msv1_p_mix(float(rng.choice([0.0, 0.3, 0.7, 0.95, 1.0])), float(rng.choice([1.5, 8.0, 40.0, 300.0])))
        key_every = int(rng.choice([0, 1, 3, 5]))
        cfg = int(rng.integers(0, 1 << 30))
        frames, keys, pal = sg.msv1_clip(cfg, w, h, n, bits=bits, p_mix=None if key_every == 1 else p_mix, key_every=key_every)
        frames = list(frames)
        for i in range(1, n):                       # mutate some frames (never frame 0: later frames need a previous one)
            r = rng.random()
            b = bytearray(frames[i])
            if r < 0.10 and b:
                b = b[: int(rng.integers(0, len(b)))]
            elif r < 0.20 and b:
                b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
            elif r < 0.25:
                b = bytearray(rng.integers(0, 256, size=int(rng.integers(0, 200)), dtype=np.uint8).tobytes())
            elif r < 0.30:
                b = b + b"\x07"
            frames[i] = bytes(b)
        mode = str(rng.choice(["host", "gpu", "gpu", "async", "staged"]))   # async: jsp_decompress_*_async / jsp_wait with the on-GPU parse; staged: the whole clip as ONE batch, replayed
        host_buffers, misalign = rng.random() < 0.15, rng.random() < 0.15
        lines = int(rng.integers(0, 60))
        depth = int(rng.choice([1, 2, 4, 8]))
        form = str(rng.choice(["one_launch_dma", "one_launch", "two_launches"]))   # how the asynchronous path runs a frame
        tag = f"{w}x{h} {bits}bit n={n} key_every={key_every} parse={mode} host={host_buffers} misalign={misalign} lines={lines} cfg={cfg}" + (f" depth={depth} form={form}" if mode == "async" else "")
        try:
            if mode == "async":
                import test_async_gpu as A
                from jsplayer_amd import MSVideo1_16bit, MSVideo1_8bit
                gpu = MSVideo1_16bit(w, h) if bits == 16 else MSVideo1_8bit(w, h, pal)
                gpu.set_option("msv1_parse", "gpu")
                gpu.set_option("msv1_async", form)
                # (a third of the asynchronous clips: the arena goes up in ranges of a few frames ahead of them, jsp_prefetch — some ranges
                # stop short of their last frame, and halfway every range is given up once)
                ranges = int(rng.integers(1, 5)) if rng.random() < 0.33 else 0
                A.drive(gpu, OracleMSVideo1(bits, w, h, pal), w, h, frames, keys, depth=depth, pinned=bool(ranges) or bool(rng.random() < 0.5), lines=lines,
                        prefetch=ranges, drop_ranges_at=(n // 2 if ranges and rng.random() < 0.5 else None))
            elif mode == "staged":
                opts = dict(msv1_parse_ahead=str(rng.choice(["on", "off"])), msv1_compact_tables=str(rng.choice(["on", "off"])), msv1_scrub_tables="1")
                tag += " " + " ".join(f"{k}={v}" for k, v in opts.items())
                drive_staged(T, bits, w, h, frames, keys, pal, lines, int(rng.integers(2, 6)), int(rng.integers(1, 4)), opts)
            else:
                drive(T, bits, w, h, frames, keys, pal, lines, mode, host_buffers, misalign)
        except AssertionError as e:
            print("BAD", tag, e, flush=True)
            bad += 1
            if bad >= 10:
                return 1
            continue
        print("ok ", tag, flush=True)
        clips += 1
        nframes += n
    print(f"fuzz finished: {clips} clips, {nframes} frames, {bad} BAD, {time.time() - t0:.0f} s, seed {seed}")
    return 1 if bad else 0


def drive(T, bits, w, h, frames, keys, pal, lines, mode, host_buffers, misalign):
    """test_msvideo1_gpu.drive_pair with the parse mode its make_gpu() applies."""
    T.PARSE_MODE = mode
    T.drive_pair(bits, w, h, frames, keys, pal, lines=lines, host=host_buffers, misalign=misalign and not host_buffers)


def drive_staged(T, bits, w, h, frames, keys, pal, lines, nbuf, replays, opts):
    """The clip as ONE staged batch into `nbuf` rotating buffers, decoded and replayed `replays` times back to back (round 6: the next replay's
    parse beside this one, compact block tables, tables poisoned before every parse) — mutated, truncated and random frames included, so that
    frames the host parser has to settle sit between the GPU-parsed ones.  Statuses, adoption, significance and every buffer against the oracle."""
    from jsplayer_amd import MSVideo1_16bit, MSVideo1_8bit
    from oracle_binding import OracleAbort, OracleMSVideo1
    n = len(frames)
    orc = OracleMSVideo1(bits, w, h, pal)
    orc.Preinit(lines)
    obufs = [np.full(w * h, 5, dtype=np.int32) for _ in range(nbuf)]
    want, where = [], []                        # per frame: None (the reference raises) or (adopted, significant or None for a key frame); the buffer it goes to
    for i in range(n):
        k = next(j for j in range(nbuf) if obufs[j] is not orc.PreviousFrame())   # the Manager's rule: any buffer but the previous frame's
        where.append(k)
        if keys[i]:
            assert orc.DecompressI(frames[i], obufs[k]) == 0
            want.append((True, None))
        else:
            try:
                data, sig = orc.DecompressP(frames[i], obufs[k])
                want.append((data is obufs[k], sig))
            except OracleAbort:
                want.append(None)
    gpu = MSVideo1_16bit(w, h) if bits == 16 else MSVideo1_8bit(w, h, pal)
    gpu.Preinit(lines)
    gpu.set_option("msv1_parse", "gpu")
    for k, v in opts.items():
        gpu.set_option(k, v)
    dsts = [T.dev_buf(w * h, 5) for _ in range(nbuf)]
    st = gpu.stage_batch(frames, [dsts[where[i]] for i in range(n)], is_key=keys)
    for run in range(2):
        for d in dsts:
            d.fill_(5)
        for _ in range(replays if run else 1):
            st.decode()
        gpu.sync()
        status, adopted, signif = st.results()
        for i in range(n):
            if want[i] is None:
                assert status[i] != 0, f"run {run} frame {i}: the reference raises here"
                continue
            assert status[i] == 0, f"run {run} frame {i}: status {status[i]}"
            if not keys[i]:
                assert bool(adopted[i]) == want[i][0], f"run {run} frame {i}: adoption"
                assert bool(signif[i]) == want[i][1], f"run {run} frame {i}: significant_changes"
        if all(x is not None for x in want):    # (a frame at which the reference raises leaves the later ones undefined there)
            for k in range(nbuf):
                assert np.array_equal(obufs[k], T.to_np(dsts[k])), f"run {run}: buffer {k} differs"
    st.close()
    gpu.StopAndClean()


if __name__ == "__main__":
    sys.exit(main())
