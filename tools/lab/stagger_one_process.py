"""lab: launch-order variants compared in ONE process on the SAME destination frames (every process gets its own luck with where its frames
lie, so runs of bench.py cannot be compared with each other).
usage: python tools/lab/stagger_one_process.py <workload> <frames: torch|pool> <value> [<value> ...]   (rounds through the list twice)
       the values go into the environment variable named by LAB_VAR (default JSP_MSV1_STAGGER) before each staging"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from jsplayer_amd import workloads as wl

name, how = sys.argv[1], sys.argv[2]
staggers = [int(a) for a in sys.argv[3:]]
W, H = wl.W, wl.H
clip = wl.build_clips(name, 0)[0]
n = len(clip.frames)
if how == "pool":
    pool = torch.empty(n * W * H, dtype=torch.int32, device="cuda")
    dsts = [pool[i * W * H:(i + 1) * W * H] for i in range(n)]
else:
    dsts = [torch.empty(W * H, dtype=torch.int32, device="cuda") for _ in range(n)]
spec = wl.WORKLOADS[name]
for rnd in range(2):
    for st in staggers:
        os.environ[os.environ.get("LAB_VAR", "JSP_MSV1_STAGGER")] = str(st)
        codec = wl.make_codec(name, clip.palette, device=0)
        staged = codec.stage_batch(clip.frames, dsts, is_key=clip.keys)
        for _ in range(3):
            staged.decode()
        codec.sync()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            staged.decode()
        codec.sync()
        dt = (time.perf_counter() - t0) / 20
        info = staged.info()
        moved = min(info["algorithmic_bytes"], info["moved_bytes"]) if "moved_bytes" in info else info["algorithmic_bytes"]
        print(f"{name} frames {how:5s} {os.environ.get('LAB_VAR', 'stagger')} {st:4d}: {dt * 1e3:.4f} ms  {moved / dt / 8e12:.4f} of 8 TB/s", flush=True)
        staged.close()
        codec.StopAndClean()
