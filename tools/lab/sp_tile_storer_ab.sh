#!/bin/bash
# Same-session A/B of the ScreenPressor key-frame launch: one wave per tile (JSP_SP_TILE_STORER=0) against workgroups of 4 / 7 resolver waves plus a
# storer wave, rings of 2 / 4 / 8 rows.  Digests verified every time.
R="${GRAFT_REPO_ROOT:-$(pwd)}"; O="$R/gpurun_out"; T="${TAG:-ab}"
cd "$R"
: > "$O/${T}_sp_tile_storer_ab.txt"
one() {  # $1 resolver waves, $2 ring rows
  JSP_SP_TILE_STORER=$1 JSP_SP_TILE_RING=$2 timeout -k 10 300 python bench.py --workload screenpressor_v4_1080p_iframes --steps 20 --warmup 3 --no-cpu-baseline --no-e2e 2>> "$O/${T}_sp_tile_storer_ab.err" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["roofline"]["frac"], d["roofline"]["step_us"], d["verified"], d["config"]["destination_frames"]["probe_GBs"])' | sed "s/^/resolvers $1 ring $2: /" | tee -a "$O/${T}_sp_tile_storer_ab.txt"
}
one 0 4
one 4 4
one 4 2
one 4 8
one 7 4
one 7 2
one 0 4
one 4 4
