#!/usr/bin/env python3
"""Writes bench_digests.json: what the CPU ORACLE decodes for the full-size (1920x1080) bench workloads of
jsplayer_amd/workloads.py, as one 64-bit truncated SHA-256 per frame.

    python tests/golden/make_bench_digests.py [--workloads a,b] [--ranks 0,1] [--merge]

bench.py compares the frames its timed kernels left in HBM with these digests after the timed region
("verified" in its JSON line) and tests/test_bench_workloads_gpu.py does the same through the staged-batch entry
points, so the kernels that are timed are checked against the oracle at the size they are timed.  Inputs are
regenerated from the seed wherever the digests are used; `stream_sha` pins them.  Like every fixture here the
digests pin the HIP path to the oracle, not the oracle to the reference (which ships no vectors).
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from jsplayer_amd import workloads as wl  # noqa: E402
from oracle_binding import OracleMSVideo1, OracleScreenPressor  # noqa: E402

OUT = os.path.join(HERE, "bench_digests.json")
ALL_RANKS = ("msvideo1_16_1080p_keyframes_m1", "screenpressor_v4_1080p_pclip300", "screenpressor_v4_1080p_iframes")   # the 8-stream configurations, and every workload bench.py's default line carries (the driver runs it on 1/2/4/8 ranks)


def oracle_clip(name, clip):
    """-> per frame: digest of the oracle's previous frame after the call, or "-" when the call left it unchanged"""
    spec = wl.WORKLOADS[name]
    orc = OracleScreenPressor(wl.W, wl.H, 24) if spec["codec"] == "sp" else OracleMSVideo1(spec["bits"], wl.W, wl.H, clip.palette)
    orc.Preinit(36)
    bufs = [np.zeros(wl.W * wl.H, np.int32) for _ in range(3)]
    out = []
    for src, key in zip(clip.frames, clip.keys):
        dst = next(b for b in bufs if b is not orc.PreviousFrame())
        if key:
            assert orc.DecompressI(src, dst) == 0
            adopted = orc.PreviousFrame() is dst
        else:
            data, _ = orc.DecompressP(src, dst)
            adopted = data is dst
        out.append(wl.digest(orc.PreviousFrame()) if adopted else "-")
    orc.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default=",".join(wl.WORKLOADS))
    ap.add_argument("--ranks", default="")
    ap.add_argument("--merge", action="store_true", help="keep the entries already in the file")
    args = ap.parse_args()
    doc = {"note": "oracle/ decodes of jsplayer_amd.workloads clips; sha256[:16] per frame; written by make_bench_digests.py",
           "digests": {}, "stream_sha": {}}
    if args.merge and os.path.exists(OUT):
        doc = json.load(open(OUT))
    for name in args.workloads.split(","):
        ranks = [int(r) for r in args.ranks.split(",")] if args.ranks else (range(8) if name in ALL_RANKS else [0])
        for rank in ranks:
            t0 = time.time()
            clips = wl.build_clips(name, rank)
            t1 = time.time()
            key = f"{name}/rank{rank}"
            doc["digests"][key] = [oracle_clip(name, c) for c in clips]
            doc["stream_sha"][key] = [hashlib.sha256(b"".join(c.frames)).hexdigest()[:16] for c in clips]
            print(f"{key}: {sum(len(c.frames) for c in clips)} frames, generate {t1 - t0:.1f} s, oracle {time.time() - t1:.1f} s", flush=True)
            with open(OUT, "w") as f:
                json.dump(doc, f, indent=0, sort_keys=True)
    print(OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
