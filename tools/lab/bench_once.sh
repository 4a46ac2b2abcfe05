#!/bin/bash
# lab: the default bench once, with its wall time; prints the headline fields
R="${GRAFT_REPO_ROOT:-$(pwd)}"; cd $R; O=gpurun_out; T="${TAG:-once}"
s=$SECONDS
timeout -k 10 500 python bench.py "$@" > $O/${T}_bench.json 2> $O/${T}_bench.err || { tail -5 $O/${T}_bench.err; exit 1; }
echo "wall $((SECONDS - s)) s"
python - "$O/${T}_bench.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]; df = d["config"]["destination_frames"]; e = d.get("e2e", {})
print(d["value"], r["frac"], (r.get("reference_fill") or {}).get("value"), df.get("allocations_tried"), df.get("probe_GBs"), e.get("value"), e.get("all_threads", {}).get("value"), e.get("batch_api", {}).get("value"))
PY
