"""Writes the small clips the ThreadSanitizer driver plays (tests/tsan/driver.cpp): MSVideo1 16/8-bit and ScreenPressor v2/v4 streams from the
product's stream generators, in one flat file:  u32 nclips | per clip: i32 kind, w, h, bpp, palette_bytes, nframes | palette | per frame: u8 key, u32 len, bytes."""
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from jsplayer_amd import streamgen as sg  # noqa: E402


def main(path):
    clips = []
    frames, keys, _ = sg.msv1_clip(71, 320, 240, 24, p_mix=sg.msv1_p_mix(0.6, 20.0), key_every=6)
    clips.append((1, 320, 240, 16, b"", frames, keys))
    frames, keys, pal = sg.msv1_clip(72, 160, 120, 16, bits=8, p_mix=sg.msv1_p_mix(0.5, 10.0), key_every=4)
    clips.append((2, 160, 120, 8, bytes(pal), frames, keys))
    chunks, keys, _ = sg.sp_clip(73, 320, 240, 16, version=4, key_every=4)
    clips.append((3, 320, 240, 24, b"", chunks, keys))
    chunks, keys, _ = sg.sp_clip(74, 100, 52, 12, version=2, key_every=3, flat_at=(5,), unchanged_at=(7,))
    clips.append((3, 100, 52, 24, b"", chunks, keys))
    with open(path, "wb") as f:
        f.write(struct.pack("<I", len(clips)))
        for kind, w, h, bpp, pal, frames, keys in clips:
            f.write(struct.pack("<6i", kind, w, h, bpp, len(pal), len(frames)))
            f.write(pal)
            for fr, k in zip(frames, keys):
                fr = bytes(fr)
                f.write(struct.pack("<BI", 1 if k else 0, len(fr)))
                f.write(fr)
    print(f"{path}: {len(clips)} clips, {sum(len(c[5]) for c in clips)} frames")


if __name__ == "__main__":
    main(sys.argv[1])
