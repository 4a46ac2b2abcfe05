"""lab: one torch tensor per frame / one torch allocation / the product's placed frame pool as destination of the same clips, in ONE process.
usage: JSP_POOL_PROBE_LOG=1 python tools/lab/pool_one_process.py [workload ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from jsplayer_amd import FramePool, workloads as wl

W, H = wl.W, wl.H
for name in sys.argv[1:] or ["msvideo1_16_1080p_keyframes_m1", "msvideo1_16_1080p_keyframes_solid"]:
    clip = wl.build_clips(name, 0)[0]
    n = len(clip.frames)
    for rnd in range(2):
        for how in ("torch", "one", "pool"):
            fp = None
            if how == "torch":
                dsts = [torch.empty(W * H, dtype=torch.int32, device="cuda") for _ in range(n)]
            elif how == "one":
                slab = torch.empty(n * W * H, dtype=torch.int32, device="cuda")
                dsts = [slab[i * W * H:(i + 1) * W * H] for i in range(n)]
            else:
                fp = FramePool(W, H, n)
                dsts = list(fp.frames)
            codec = wl.make_codec(name, clip.palette, device=0)
            staged = codec.stage_batch(clip.frames, dsts, is_key=clip.keys)
            for _ in range(3):
                staged.decode()
            codec.sync()
            t0 = time.perf_counter()
            for _ in range(20):
                staged.decode()
            codec.sync()
            dt = (time.perf_counter() - t0) / 20
            info = staged.info()
            moved = min(info["algorithmic_bytes"], info["moved_bytes"])
            extra = f"  (pool: {fp.attempts} tried, probe {fp.store_rate:.0f} GB/s)" if fp else ""
            print(f"{name} frames {how:5s}: {dt * 1e3:.4f} ms  {moved / dt / 8e12:.4f} of 8 TB/s{extra}", flush=True)
            staged.close()
            codec.StopAndClean()
            del dsts
            if fp:
                fp.close()
            torch.cuda.empty_cache()
