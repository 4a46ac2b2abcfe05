#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes of one bench.py command (FETCH_SIZE, WRITE_SIZE; separate runs as the MI355X
guide prescribes) into HBM bytes PER BENCH STEP for the kernels that workload launches.

    tools/pmc_summary.py <fetch_dir> <write_dir> <bench json line of the same workload> <out.json>
    tools/pmc_summary.py --install <out.json> <profiles/name.json>     # copy under profiles/ and register it in
                                                                       # profiles/traffic_by_workload.json (bench.py)

gfx950 corrections (MI355X_MICROARCH.md §HBM): counters are in KiB; FETCH_SIZE reports exactly half of the bytes
of a wide coalesced streaming read, so it is doubled; WRITE_SIZE is exact for 16-byte-per-lane streaming stores.
Per kernel name the counter is averaged over its dispatches and multiplied by that kernel's launches per step
(dispatches / steps the profiled command ran), then summed over the workload's kernels.
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(d, counter, names):
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            for n in names:
                if n in r["Kernel_Name"]:
                    acc.setdefault(n, []).append(float(r["Counter_Value"]))
                    break
    return acc


def summarise(fetch_dir, write_dir, bench_json, out_path):
    line = [l for l in open(bench_json) if l.strip().startswith("{")][-1]
    b = json.loads(line)
    names = [k.strip() for part in b["roofline"]["kernel"].split("|") for k in part.split("+")]
    names = sorted(set(names), key=len, reverse=True)     # longest first: msv1_blocks_temporal_kernel before msv1_blocks_kernel
    steps_run = None
    out = {"workload": b["config"]["workload"], "kernels": b["roofline"]["kernel"], "frames_per_step": b["config"]["frames_per_step"],
           "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950: FETCH_SIZE counts 128-B requests as 64 B)",
           "per_kernel": {}}
    fetch, write = per_kernel(fetch_dir, "FETCH_SIZE", names), per_kernel(write_dir, "WRITE_SIZE", names)
    rd = wr = 0.0
    for n in names:
        if n not in fetch or n not in write:
            raise SystemExit(f"no counter rows for {n}")
        # the PMC passes ran `--warmup 1 --steps 2`: 3 steps; a kernel also launched once at staging shows 4 dispatches
        steps_run = 3
        per_step = max(1, round(len(fetch[n]) / steps_run))
        f_b, w_b = 2 * 1024 * sum(fetch[n]) / len(fetch[n]), 1024 * sum(write[n]) / len(write[n])
        out["per_kernel"][n] = {"dispatches_seen": [len(fetch[n]), len(write[n])], "launches_per_step": per_step,
                                "read_bytes_per_launch": f_b, "write_bytes_per_launch": w_b}
        rd += f_b * per_step
        wr += w_b * per_step
    sys.path.insert(0, ROOT)
    from jsplayer_amd.workloads import kernel_source_digest
    out["kernel_sources"] = kernel_source_digest(b["roofline"]["kernel"])      # (of the tree that ran: this script runs in the same gpurun call)
    out.update({"hbm_read_bytes": rd, "hbm_write_bytes": wr, "hbm_bytes": rd + wr,
                "algorithmic_bytes_per_step": b["roofline"]["algorithmic_bytes_per_step"],
                "moved_bytes_per_step": b["roofline"]["moved_bytes_per_step"]})
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps(out, indent=1))


def install(src, dst):
    doc = json.load(open(src))
    dst_abs = dst if os.path.isabs(dst) else os.path.join(ROOT, dst)
    json.dump(doc, open(dst_abs, "w"), indent=1)
    reg_path = os.path.join(ROOT, "profiles", "traffic_by_workload.json")
    reg = {"note": "HBM bytes per bench step from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/pmc_summary.py); bench.py "
                   "copies an entry into roofline.traffic only while its kernel list, frames_per_step and the digest of the kernels' source files (workloads.kernel_source_digest) still match", "per_step": {}}
    if os.path.exists(reg_path):
        old = json.load(open(reg_path))
        if "per_step" in old:
            reg = old
    reg["per_step"][doc["workload"]] = {"kernels": doc["kernels"], "frames_per_step": doc["frames_per_step"],
                                        "hbm_bytes": doc["hbm_bytes"], "source": os.path.relpath(dst_abs, ROOT),
                                        "kernel_sources": doc.get("kernel_sources")}
    json.dump(reg, open(reg_path, "w"), indent=1)
    print("registered", doc["workload"], "->", os.path.relpath(dst_abs, ROOT))


if __name__ == "__main__":
    if sys.argv[1] == "--install":
        install(sys.argv[2], sys.argv[3])
    else:
        summarise(*sys.argv[1:5])
