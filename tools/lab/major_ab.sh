#!/bin/bash
# lab: launch order of the fused kernel's tiles — tile-major over runs of G frames (JSP_MSV1_TILE_MAJOR_FRAMES), one clip per step
R="${GRAFT_REPO_ROOT:-$(pwd)}"; cd $R
export JSP_BENCH_CLIPS=1
for r in $(seq 1 ${ROUNDS:-2}); do
 for g in ${GS:-0 1 8 32 128}; do
  printf "major frames %4d: " $g
  JSP_MSV1_TILE_MAJOR_FRAMES=$g timeout -k 10 300 python bench.py --workload ${WORKLOAD:-msvideo1_16_1080p_keyframes_m1} --steps 20 --warmup 3 --no-e2e --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], d.get('verified'), r['frac'], r['measured_ceiling']['value'])"
 done
done
