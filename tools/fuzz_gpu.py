#!/usr/bin/env python3
"""Randomised GPU campaign for the ScreenPressor kernels: random geometry, stream version, band height, clip
structure (key frames, flat frames, unchanged frames, light / heavy motion, noise), buffer reuse and alignment;
every adopted frame of a staged batch must equal the image the encoder was given.

    python tools/fuzz_gpu.py [seconds] [seed]        (prints one line per clip and a summary; exit 1 on mismatch)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    import torch
    from jsplayer_amd import ScreenPressor
    from jsplayer_amd import streamgen as sg
    rng = np.random.default_rng(seed)
    t0, clips, frames_checked, skipped, damaged, async_clips = time.time(), 0, 0, 0, 0, 0
    while time.time() - t0 < budget:
        w = int(rng.choice([int(rng.integers(4, 160)) * 4, int(rng.integers(17, 700)), int(rng.integers(64, 600)) * 4]))
        h = int(rng.integers(9, 200))
        version = int(rng.choice([2, 3, 4]))
        bpp = int(rng.choice([24, 24, 16]))
        n = int(rng.integers(3, 11))
        key_every = int(rng.choice([0, 1, 2, 4]))
        unchanged = tuple(int(x) for x in rng.choice(np.arange(1, n), size=min(2, n - 1), replace=False)) if rng.random() < 0.5 else ()
        flat = (int(rng.integers(1, n)),) if rng.random() < 0.3 else ()
        mix = {}
        for i in range(1, n):
            if rng.random() < 0.3:
                m = float(rng.choice([0.1, 0.3, 0.6, 0.9]))
                mix[i] = dict(unchanged=float(rng.uniform(0, 1 - m)), motion=m)
        noise = float(rng.choice([0.0, 0.05, 0.3, 0.9]))
        band = str(rng.choice(["auto", "0", str(int(rng.integers(1, 60)))]))
        cfg = int(rng.integers(0, 1 << 30))
        try:
            chunks, keys, frames = sg.sp_clip(cfg, w, h, n, bpp=bpp, version=version, key_every=key_every, unchanged_at=unchanged,
                                              flat_at=flat, p_mix_at=mix, noise=noise)
        except RuntimeError as e:      # content the version-3 model cannot code (see sp_encoder.cpp): not a decoder matter
            print(f"skip {w}x{h} v{version} noise={noise} cfg={cfg}: {str(e)[:60]}", flush=True)
            skipped += 1
            continue
        gpu = ScreenPressor(w, h, bpp)
        gpu.Preinit(int(rng.integers(0, 60)))
        gpu.set_option("sp_band_rows", band)
        if rng.random() < 0.2:
            gpu.set_option("sp_inter_fusion", "off")
        misalign = rng.random() < 0.2
        def buf():
            if misalign:
                return torch.full((w * h + 4,), -1, dtype=torch.int32, device="cuda")[1:1 + w * h]
            return torch.full((w * h,), -1, dtype=torch.int32, device="cuda")
        dsts = [buf() for _ in range(n)]
        st = gpu.stage_batch(chunks, dsts, is_key=keys)
        st.decode()
        gpu.sync()
        status, adopted, _ = st.results()
        ok = status == [0] * n
        for i in range(n):
            if adopted[i] and not np.array_equal(dsts[i].cpu().numpy().view(np.uint32), frames[i]):
                ok = False
                print(f"MISMATCH frame {i}", flush=True)
        print(f"{'ok ' if ok else 'BAD'} {w}x{h} v{version} bpp{bpp} n={n} key_every={key_every} unchanged={unchanged} flat={flat} "
              f"mix={sorted(mix)} noise={noise} band={band} misalign={misalign} cfg={cfg}", flush=True)
        st.close()
        gpu.StopAndClean()
        if not ok:
            return 1
        clips += 1
        frames_checked += sum(adopted)
        # ---- the same clip with damaged frames through the per-call API: no parity claim on garbage (the reference
        # raises, spins or paints noise), but nothing may fault and a fresh key frame must decode exactly again
        if rng.random() < 0.3 and keys[0]:
            from jsplayer_amd import CodecError
            gpu = ScreenPressor(w, h, bpp)
            gpu.Preinit(36)
            bufs = [buf() for _ in range(3)]
            pick = lambda: next(b for b in bufs if b is not gpu.PreviousFrame())
            gpu.DecompressI(chunks[0], pick())
            for k in range(12):
                i = int(rng.integers(0, n))
                b = bytearray(chunks[i])
                r = rng.random()
                if r < 0.35 and len(b) > 1:
                    b = b[: int(rng.integers(1, len(b)))]
                elif r < 0.7 and len(b) > 1:
                    for _ in range(int(rng.integers(1, 4))):
                        b[int(rng.integers(1, len(b)))] ^= int(rng.integers(1, 256))
                else:
                    b = bytearray([b[0] if b else 0x32]) + bytearray(rng.integers(0, 256, size=int(rng.integers(0, 500)), dtype=np.uint8).tobytes())
                try:
                    (gpu.DecompressI if keys[i] else gpu.DecompressP)(bytes(b), pick())
                except CodecError:
                    pass
                damaged += 1
            dst = pick()
            if int(gpu.DecompressI(chunks[0], dst)) != 0 or not np.array_equal(dst.cpu().numpy().view(np.uint32), frames[0]):
                print("BAD: no exact recovery after damaged frames", w, h, version, cfg, flush=True)
                return 1
            gpu.StopAndClean()
        # ---- the asynchronous calls (worker threads, 8 frames in flight, a buffer per frame) against the synchronous ones on the
        # same frames, some of them cut short or with a byte flipped: where a frame adopts nothing against what its first byte
        # promised, the frames behind it must still come out as the one-frame-at-a-time path leaves them — states, errors,
        # previous-frame identities and the pixels of every buffer
        if rng.random() < 0.4 and not misalign:
            from jsplayer_amd import CodecError
            seq = [bytes(c) for c in chunks]
            for _ in range(int(rng.integers(0, 3))):
                i = int(rng.integers(0, n))
                b = bytearray(seq[i])
                if rng.random() < 0.6 and len(b) > 2:
                    b = b[: int(rng.integers(1, len(b)))]
                elif len(b) > 2:
                    b[int(rng.integers(1, len(b)))] ^= int(rng.integers(1, 256))
                seq[i] = bytes(b)
            threads = str(rng.choice(["1", "2", "4"]))
            runs = []
            for mode in ("sync", "async"):
                gpu = ScreenPressor(w, h, bpp)
                gpu.Preinit(36)
                gpu.set_option("sp_async_threads", threads)
                gpu.set_option("async_depth", "8")
                bufs = [torch.full((w * h,), 7, dtype=torch.int32, device="cuda") for _ in range(n)]
                log, tickets = [], []

                def settle(i, fn):
                    try:
                        r = fn()
                        out = ("state", int(r)) if keys[i] else ("p", next((k for k in range(n) if bufs[k] is r.data_pnt), None), r.significant_changes)
                    except CodecError:
                        out = ("raise",)
                    log.append((i, out))
                for i, (c, k) in enumerate(zip(seq, keys)):
                    if mode == "sync":
                        settle(i, (lambda: gpu.DecompressI(c, bufs[i])) if k else (lambda: gpu.DecompressP(c, bufs[i])))
                    else:
                        if len(tickets) == 8:
                            j, t = tickets.pop(0)
                            settle(j, lambda: gpu.wait(t))
                        tickets.append((i, (gpu.DecompressI_async if k else gpu.DecompressP_async)(c, bufs[i])))
                for j, t in tickets:
                    settle(j, lambda: gpu.wait(t))
                torch.cuda.synchronize()
                prev = gpu.PreviousFrame()
                runs.append((log, next((k for k in range(n) if bufs[k] is prev), None), [x.cpu().numpy() for x in bufs]))
                gpu.StopAndClean()
            same = runs[0][0] == runs[1][0] and runs[0][1] == runs[1][1] and all(np.array_equal(a, b_) for a, b_ in zip(runs[0][2], runs[1][2]))
            if not same:
                print(f"BAD: asynchronous calls differ from the synchronous ones: {w}x{h} v{version} bpp{bpp} cfg={cfg} threads={threads}", runs[0][0], runs[1][0], flush=True)
                return 1
            async_clips += 1
    print(f"fuzz finished: {clips} clips, {frames_checked} frames, {skipped} skipped as unencodable, {damaged} damaged frames survived, {async_clips} clips through the asynchronous calls against the synchronous ones, {time.time() - t0:.0f} s, seed {seed}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
