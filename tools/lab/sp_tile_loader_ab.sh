#!/bin/bash
# Same-session A/B of the ScreenPressor key-frame launch: one wave per tile fetching its own record windows (JSP_SP_TILE_LOADER=0) against
# workgroups of 4 / 7 tile waves plus a loader wave, with 2 / 3 window buffers per tile and a limit on the row stores a tile wave keeps in flight.
# Digests verified every time.
R="${GRAFT_REPO_ROOT:-$(pwd)}"; O="$R/gpurun_out"; T="${TAG:-ab}"
cd "$R"
: > "$O/${T}_sp_tile_loader_ab.txt"
one() {  # $1 loader, $2 nbuf, $3 vmcnt
  JSP_SP_TILE_LOADER=$1 JSP_SP_TILE_NBUF=$2 JSP_SP_TILE_VMCNT=$3 timeout -k 10 300 python bench.py --workload screenpressor_v4_1080p_iframes --steps 20 --warmup 3 --no-cpu-baseline --no-e2e 2>> "$O/${T}_sp_tile_loader_ab.err" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["step_us"], d["verified"], d["config"]["destination_frames"]["probe_GBs"])' | sed "s/^/loader $1 nbuf $2 store limit $3: /" | tee -a "$O/${T}_sp_tile_loader_ab.txt"
}
one 0 2 0
one 4 2 0
one 4 3 0
one 4 2 2
one 4 2 4
one 4 3 4
one 4 3 8
one 7 2 0
one 7 2 4
one 0 2 0
one 4 2 0
