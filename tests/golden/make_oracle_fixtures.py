"""Writes oracle_fixtures.npz: compressed streams + the frames / flags the CPU ORACLE produces for them.

These pin the oracle (and through it the HIP path) against regressions and travel to the GPU box;
they do NOT pin the oracle against the reference, which ships no vectors.  Streams
come from jsplayer_amd.streamgen (seeded); frames are stored for the small sizes, SHA-256 digests
for 1920x1080.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from jsplayer_amd import streamgen as sg  # noqa: E402
from oracle_binding import OracleMSVideo1, OracleScreenPressor  # noqa: E402


def decode(orc, w, h, chunks, keys):
    orc.Preinit(36)
    bufs = [np.zeros(w * h, np.int32) for _ in range(3)]
    frames, flags = [], []
    for c, k in zip(chunks, keys):
        dst = next(b for b in bufs if b is not orc.PreviousFrame())
        if k:
            assert orc.DecompressI(c, dst) == 0
            flags.append((1, 0))
        else:
            data, sig = orc.DecompressP(c, dst)
            flags.append((int(data is dst), int(sig)))
        frames.append(orc.PreviousFrame().copy())
    return frames, flags


def main():
    out = {}
    cases = []
    for bits in (16, 8):
        for (w, h) in [(16, 16), (64, 48), (320, 240), (1920, 1080)]:
            n = 3 if w >= 320 else 6
            chunks, keys, pal = sg.msv1_clip(7000 + bits, w, h, n, bits=bits, p_mix=sg.msv1_p_mix(0.7, 20.0))
            frames, flags = decode(OracleMSVideo1(bits, w, h, pal), w, h, chunks, keys)
            cases.append((f"msv1_{bits}_{w}x{h}", w, h, chunks, keys, frames, flags, pal or b""))
    for version in (2, 3, 4):
        for (w, h) in [(64, 48), (320, 240), (1920, 1080)]:
            n = 2 if w >= 1920 else 5
            chunks, keys, _ = sg.sp_clip(7100 + version, w, h, n, version=version, flat_at=(3,) if n > 3 else ())
            frames, flags = decode(OracleScreenPressor(w, h, 24), w, h, chunks, keys)
            cases.append((f"sp_v{version}_{w}x{h}", w, h, chunks, keys, frames, flags, b""))
    names = []
    for name, w, h, chunks, keys, frames, flags, pal in cases:
        names.append(name)
        out[name + "/shape"] = np.array([w, h], dtype=np.int32)
        out[name + "/keys"] = np.array(keys, dtype=np.uint8)
        out[name + "/flags"] = np.array(flags, dtype=np.uint8)
        out[name + "/palette"] = np.frombuffer(bytes(pal), dtype=np.uint8)
        out[name + "/lens"] = np.array([len(c) for c in chunks], dtype=np.int64)
        big = w * h > 100000
        if big:   # inputs are regenerated from the seed; only digests are stored
            out[name + "/stream_sha256"] = np.array([hashlib.sha256(c).hexdigest() for c in chunks])
            out[name + "/frame_sha256"] = np.array([hashlib.sha256(f.tobytes()).hexdigest() for f in frames])
        else:
            out[name + "/stream"] = np.frombuffer(b"".join(chunks), dtype=np.uint8)
            out[name + "/frames"] = np.stack(frames)
    out["names"] = np.array(names)
    path = os.path.join(HERE, "oracle_fixtures.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes,", len(names), "cases")


if __name__ == "__main__":
    main()
