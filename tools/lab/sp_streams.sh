#!/bin/bash
# lab: ScreenPressor key frames through the asynchronous calls, 1 / 4 / 16 streams on ONE file (round 2's e2e) — where does the rate go?
R="${GRAFT_REPO_ROOT:-$(pwd)}"; cd $R
python tools/write_workload_avi.py screenpressor_v4_1080p_iframes 64 /tmp/sp64.avi
for s in 1 4 8 16; do
  for rep in 1 2; do
    printf "streams %2d: " $s
    ./examples/jsp_play /tmp/sp64.avi --pipelined --quiet --depth 8 --streams $s --repeat 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d.get('mpixels_per_s', d.get('value', 0))), {k: d[k] for k in d if k in ('frames','ms_per_frame','streams')})"
  done
done
