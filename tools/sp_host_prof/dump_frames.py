"""Writes the compressed frames of a bench workload's first clip to a flat file for tools/sp_host_prof/prof.cpp:
u32 count, then per frame u32 length, u8 key flag, the bytes."""
import struct
import sys

from jsplayer_amd import workloads as wl

name, frames, path = sys.argv[1], int(sys.argv[2]), sys.argv[3]
clip = wl.build_clips(name, 0, frames=frames)[0]
with open(path, "wb") as f:
    f.write(struct.pack("<I", len(clip.frames)))
    for b, k in zip(clip.frames, clip.keys):
        f.write(struct.pack("<IB", len(b), 1 if k else 0))
        f.write(bytes(b))
