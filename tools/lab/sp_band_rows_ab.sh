#!/bin/bash
# lab: ScreenPressor key frames rebuilt in bands of 90 (auto) / 60 / 45 / 30 rows (option sp_band_rows), alternating, one call.  step ms | frac | verified | pool probe
R="${GRAFT_REPO_ROOT:-$(pwd)}"
one() { (cd $R && JSP_BENCH_OPTIONS="sp_band_rows=$1" python bench.py --workload screenpressor_v4_1080p_iframes --steps 30 --warmup 5 --no-e2e --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['verified'], d['config']['destination_frames']['probe_GBs'], d['roofline']['moved_bytes_per_step'])"); }
for i in 1 2; do for b in auto 60 45 30; do echo -n "band $b: "; one $b; done; done
