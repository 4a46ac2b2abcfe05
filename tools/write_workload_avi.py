"""Writes the first clip of a bench workload as an AVI file (what bench.py's e2e leg plays): for profiling examples/jsp_play.
usage: python tools/write_workload_avi.py <workload> <frames> <out.avi>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jsplayer_amd import avi, workloads as wl

name, n, path = sys.argv[1], int(sys.argv[2]), sys.argv[3]
spec = wl.WORKLOADS[name]
clip = wl.build_clips(name, 0, frames=n)[0]
W, H = 1920, 1080
blob = avi.write_avi(W, H, clip.frames[:n], fourcc=b"SCPR" if spec["codec"] == "sp" else b"CRAM",
                     bpp=24 if spec["codec"] == "sp" else spec["bits"], palette=clip.palette, key_flags=clip.keys[:n])
open(path, "wb").write(blob)
