#!/usr/bin/env python3
"""Per-call latency of the drop-in entry points (one DecompressI/DecompressP per frame, synchronous,
stream bytes coming from host memory every call) — the PCIe-/host-inclusive numbers DESIGN.md quotes
next to the resident-input throughput of bench.py."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from jsplayer_amd import MSVideo1_16bit, ScreenPressor
from jsplayer_amd import streamgen as sg

w, h = 1920, 1080
def run(name, codec, chunks, keys, reps=3):
    bufs = [torch.empty(w * h, dtype=torch.int32, device="cuda") for _ in range(3)]
    codec.Preinit(36)
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        for c, k in zip(chunks, keys):
            dst = next(b for b in bufs if b is not codec.PreviousFrame())
            if k: codec.DecompressI(c, dst)
            else: codec.DecompressP(c, dst)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    n = len(chunks)
    print(f"{name:55s} {best / n * 1e3:8.3f} ms/frame  {n * w * h / best / 1e6:10.0f} Mpx/s")

frames, keys, _ = sg.msv1_clip(2, w, h, 32)
c = MSVideo1_16bit(w, h); c.set_option("msv1_parse", "host"); run("MSVideo1 key frames, host parse, per-frame calls", c, frames, [True] * 32)
c = MSVideo1_16bit(w, h); c.set_option("msv1_parse", "gpu"); run("MSVideo1 key frames, on-GPU parse, per-frame calls", c, frames, [True] * 32)
frames, keys, _ = sg.msv1_clip(2, w, h, 32, p_mix=sg.msv1_p_mix(0.7, 40.0))
c = MSVideo1_16bit(w, h); c.set_option("msv1_parse", "host"); run("MSVideo1 70% skipped inter frames, host parse", c, frames, keys)
c = MSVideo1_16bit(w, h); c.set_option("msv1_parse", "gpu"); run("MSVideo1 70% skipped inter frames, on-GPU parse", c, frames, keys)
chunks, keys, _ = sg.sp_clip(4, w, h, 24, version=4)
c = ScreenPressor(w, h, 24); run("ScreenPressor v4 clip (1 key + 23 inter), per-frame calls", c, chunks, keys, reps=1)
host = [np.empty(w * h, dtype=np.int32) for _ in range(3)]
c = MSVideo1_16bit(w, h); c.Preinit(36)
t0 = time.perf_counter()
for f in frames[:16]:
    dst = next(b for b in host if b is not c.PreviousFrame())
    c.DecompressP(f, dst) if c.PreviousFrame() is not None else c.DecompressI(f, dst)
dt = time.perf_counter() - t0
print(f"{'MSVideo1 inter frames, HOST frame buffers (compat mode)':55s} {dt / 16 * 1e3:8.3f} ms/frame  {16 * w * h / dt / 1e6:10.0f} Mpx/s")
