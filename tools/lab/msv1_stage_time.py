"""lab: jsp_stage_batch + jsp_staged_decode of N MSVideo1 M1 key frames FROM HOST BYTES, wall clock: first call (buffers are
allocated) and re-staging into the same batch object; sources in pageable memory and in pinned memory (jsp_host_alloc)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import torch
from jsplayer_amd import workloads as wl
from jsplayer_amd.codec import HostBuffer

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
name = "msvideo1_16_1080p_keyframes_m1"
clip = wl.build_clips(name, 0, frames=n)[0]
W, H = wl.W, wl.H
px = n * W * H
dsts = [torch.empty(W * H, dtype=torch.int32, device="cuda") for _ in range(n)]
total = sum(len(f) for f in clip.frames)
hb = HostBuffer(total + 64)
pinned, at = [], 0
for f in clip.frames:
    hb.array[at:at + len(f)] = np.frombuffer(f, dtype=np.uint8)
    pinned.append(hb.array[at:at + len(f)])
    at += len(f)
for label, srcs in (("pageable", clip.frames), ("pinned", pinned)):
    codec = wl.make_codec(name, clip.palette)
    torch.cuda.synchronize()
    st = None
    for rep in range(4):
        t0 = time.perf_counter()
        st = codec.stage_batch(srcs, dsts, is_key=clip.keys, reuse=st)
        t1 = time.perf_counter()
        st.decode()
        codec.sync()
        t2 = time.perf_counter()
        status, adopted, _ = st.results()
        i = st.info()
        print(f"{label:9s} rep {rep}: stage {1e3*(t1-t0):7.1f} ms  decode+sync {1e3*(t2-t1):6.1f} ms  -> {px/(t2-t0)/1e9:6.1f} Gpx/s | host_stage {i['host_stage_ms']:.1f} h2d {i['h2d_ms']:.1f} device_parse {i['device_parse_ms']:.1f} | ok {not any(status)}", flush=True)
    st.close()
    codec.StopAndClean()
