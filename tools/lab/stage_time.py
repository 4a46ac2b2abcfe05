import sys, time
sys.path.insert(0, '.')
import torch
from jsplayer_amd import workloads as wl
name = "screenpressor_v4_1080p_iframes"
clips = wl.build_clips(name, 0, frames=64)
for threads in ("1", "4", "8", "16"):
    t0 = time.time()
    work = wl.StagedWorkload(name, clips, options={"sp_host_threads": threads})
    work.step(); work.sync()
    dt = time.time() - t0
    info = work.infos[0]
    gold = wl.golden_digests(name, 0)
    bad = work.mismatches([gold[0][:64]]) if gold else None
    print(f"threads {threads}: stage+decode {dt*1e3:.0f} ms for {info['frames']} frames, host_stage_ms {info.get('host_stage_ms')}, mismatches {bad}")
    work.close()
