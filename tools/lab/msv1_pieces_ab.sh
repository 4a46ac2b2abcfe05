#!/bin/bash
# Same-session A/B of the MSVideo1 inter-frame batch: the table-writing parse as ONE launch in front of the temporal launch (JSP_MSV1_PARSE_PIECES=1)
# against 2 / 4 / 8 pieces on a side stream next to the temporal launches of the pieces before.  Digests verified every time.
R="${GRAFT_REPO_ROOT:-$(pwd)}"; O="$R/gpurun_out"; T="${TAG:-ab}"
cd "$R"
: > "$O/${T}_msv1_pieces_ab.txt"
for round in 1 2; do
  for p in 1 2 4 8 16; do
    JSP_MSV1_PARSE_PIECES=$p timeout -k 10 300 python bench.py --workload msvideo1_16_1080p_inter70 --steps 20 --warmup 3 --no-cpu-baseline --no-e2e 2>> "$O/${T}_msv1_pieces_ab.err" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["step_us"], d["roofline"]["launches_per_step"], d["verified"], d["lookback_fallbacks"])' | sed "s/^/pieces $p: /" | tee -a "$O/${T}_msv1_pieces_ab.txt"
  done
done
