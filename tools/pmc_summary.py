#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as the MI355X guide
prescribes) into profiles/<tag>_traffic.json and profiles/traffic_latest.json.

gfx950 corrections (MI355X_MICROARCH.md §HBM): counters are in KiB; FETCH_SIZE reports exactly
half of the bytes of a wide coalesced streaming read, so it is doubled; WRITE_SIZE is exact for
16-byte-per-lane streaming stores.

    tools/pmc_summary.py <fetch_dir> <write_dir> <kernel-substring> <workload> <tag>
"""
import csv
import glob
import json
import os
import sys


def mean_counter(d, counter, kernel):
    vals = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"]))
    if not vals:
        raise SystemExit(f"no {counter} rows for {kernel} under {d}")
    return sum(vals) / len(vals), len(vals)


def main():
    fetch_dir, write_dir, kernel, workload, tag = sys.argv[1:6]
    fetch_kib, nf = mean_counter(fetch_dir, "FETCH_SIZE", kernel)
    write_kib, nw = mean_counter(write_dir, "WRITE_SIZE", kernel)
    out = {
        "workload": workload,
        "kernel": kernel,
        "fetch_size_kib_raw": fetch_kib,
        "write_size_kib_raw": write_kib,
        "dispatches_averaged": [nf, nw],
        "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950: FETCH_SIZE counts 128-B requests as 64 B)",
        "hbm_read_bytes_per_launch": 2 * fetch_kib * 1024,
        "hbm_write_bytes_per_launch": write_kib * 1024,
        "hbm_bytes_per_launch": (2 * fetch_kib + write_kib) * 1024,
    }
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # tag "scratch": print only (the caller redirects); otherwise also refresh what bench.py reads
    for name in (() if tag == "scratch" else (f"{tag}_traffic.json", "traffic_latest.json")):
        json.dump(out, open(os.path.join(root, "profiles", name), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
