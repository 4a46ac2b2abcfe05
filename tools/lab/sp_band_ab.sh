#!/bin/bash
# Same-session A/B of the ScreenPressor key-frame launch over the band height (rows per tile; one wave per tile): shorter bands = more,
# shorter-lived waves (better balance at the end of the launch) against more seed rows.  Digests verified every time.
set -eo pipefail
R="${GRAFT_REPO_ROOT:-$(pwd)}"; O="$R/gpurun_out"; T="${TAG:-ab}"
cd "$R"
: > "$O/${T}_sp_band_ab.jsonl"
for round in 1 2; do
  for b in auto 60 45 36 30 24; do
    JSP_BENCH_OPTIONS="sp_band_rows=$b" timeout -k 10 300 python bench.py --workload screenpressor_v4_1080p_iframes --steps 20 --warmup 3 --no-cpu-baseline --no-e2e >> "$O/${T}_sp_band_ab.jsonl" 2>> "$O/${T}_sp_band_ab.err"
    echo "band $b: $(tail -1 "$O/${T}_sp_band_ab.jsonl" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["roofline"]["frac"], d["roofline"]["step_us"], d["verified"], d["config"]["input_bytes_per_step"], d["config"]["destination_frames"]["probe_GBs"])')"
  done
done
