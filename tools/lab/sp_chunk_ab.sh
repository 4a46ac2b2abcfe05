#!/bin/bash
# Same-session A/B of the ScreenPressor inter-frame group launch: the loader-wave kernel (JSP_SP_GROUP_CHUNK=0: a workgroup walks
# the whole group) against the time-split kernel with 4 / 8 / 16 frames per workgroup, twice round, digests verified every time.
set -eo pipefail
R="${GRAFT_REPO_ROOT:-$(pwd)}"; O="$R/gpurun_out"; T="${TAG:-ab}"
cd "$R"
: > "$O/${T}_sp_chunk_ab.jsonl"
for round in 1 2; do
  for c in 0 4 8 16; do
    JSP_SP_GROUP_CHUNK=$c timeout -k 10 300 python bench.py --workload screenpressor_v4_1080p_pclip300 --steps 20 --warmup 3 --no-cpu-baseline --no-e2e >> "$O/${T}_sp_chunk_ab.jsonl" 2>> "$O/${T}_sp_chunk_ab.err"
    echo "chunk $c: $(tail -1 "$O/${T}_sp_chunk_ab.jsonl" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["step_us"], d["verified"], d["config"]["destination_frames"]["probe_GBs"])')"
  done
done
