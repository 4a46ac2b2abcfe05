#!/usr/bin/env python3
"""One rocprofv3 --kernel-trace --stats run of bench.py and the un-profiled run of the same command FROM THE SAME gpurun CALL, side by side:

    tools/profile_pair.py <kernel_stats.csv> <bench line under rocprof> <bench line without the profiler> <out.json>

For every kernel the workload names: launches, average duration, launches per step; the step time the profile implies (sum of
avg x launches per step) against the step times the two bench lines measured with their own HIP events; the roofline fraction
each implies.  The profile's kernel time per step cannot exceed the step time of the run it was taken from; how far the
un-profiled run of the same call is from it is what the profiler (and another draw of the frame pool's placement) cost."""
import csv
import json
import sys


def last_json(path):
    return json.loads([l for l in open(path) if l.strip().startswith("{")][-1])


def main():
    stats, under, plain, out_path = sys.argv[1:5]
    u, p = last_json(under), last_json(plain)
    names = sorted({k.strip() for part in u["roofline"]["kernel"].split("|") for k in part.split("+")}, key=len, reverse=True)
    steps_run = u["steps"] + u["warmup"] + 1            # warm-up + timed + the verification replay (bench.py)
    rows = {}
    for r in csv.DictReader(open(stats)):
        for n in names:
            if n in r["Name"]:
                e = rows.setdefault(n, {"calls": 0, "total_ns": 0.0})
                e["calls"] += int(r["Calls"])
                e["total_ns"] += float(r["TotalDurationNs"])
                break
    kernels, implied = {}, 0.0
    for n, e in rows.items():
        avg_us = e["total_ns"] / e["calls"] / 1e3
        per_step = max(1, round(e["calls"] / steps_run))
        kernels[n] = {"calls": e["calls"], "avg_us": round(avg_us, 2), "launches_per_step": per_step}
        implied += avg_us * per_step
    counted = min(u["roofline"]["algorithmic_bytes_per_step"], u["roofline"]["moved_bytes_per_step"])
    doc = {
        "workload": u["config"]["workload"],
        "kernels": kernels,
        "step_us_implied_by_profile": round(implied, 2),
        "under_rocprof": {"step_us": u["roofline"]["step_us"], "frac": u["roofline"]["frac"], "pool_probe_GBs": u["config"]["destination_frames"].get("probe_GBs")},
        "same_call_without_profiler": {"step_us": p["roofline"]["step_us"], "frac": p["roofline"]["frac"], "pool_probe_GBs": p["config"]["destination_frames"].get("probe_GBs")},
        "frac_implied_by_profile": round(counted / (implied * 1e-6) / 1e9 / 8000.0, 4) if implied else None,
        # kernels of one stream: their durations add up to at most the step.  Kernels on two streams side by side (MSVideo1 inter-frame batches since round 6:
        # the next replay's table-writing parse beside this replay's temporal launch, option msv1_parse_ahead) overlap: there the longest launch bounds the step.
        "launches_overlap": len(kernels) > 1 and implied > u["roofline"]["step_us"] * 1.01,
        "consistent": implied <= u["roofline"]["step_us"] * 1.01 or
                      (len(kernels) > 1 and max(k["avg_us"] * k["launches_per_step"] for k in kernels.values()) <= u["roofline"]["step_us"] * 1.01),
        "note": "kernel time per step from rocprofv3's kernel trace against the HIP-event step time of the SAME process (under_rocprof) and of an "
                "un-profiled run of the same command in the same gpurun call (a process of its own: its frame pools are placed anew)",
    }
    json.dump(doc, open(out_path, "w"), indent=1)
    print(json.dumps(doc))


if __name__ == "__main__":
    main()
