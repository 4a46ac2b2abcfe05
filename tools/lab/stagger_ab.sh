#!/bin/bash
# lab: frames of a batch start their tiles (JSP_MSV1_STAGGER) rounds apart in the tile-major launch order, frames in one torch allocation (the reliably slow placement) / one tensor each
R="${GRAFT_REPO_ROOT:-$(pwd)}"; cd $R
export JSP_BENCH_CLIPS=1
for w in ${WORKLOADS:-msvideo1_16_1080p_keyframes_solid msvideo1_16_1080p_keyframes_m1}; do
 for pool in ${POOLS:-1 torch}; do
  for st in ${STS:-0 2 4 8 16 32}; do
   printf "%-36s frames %-6s stagger %3d: " $w $pool $st
   JSP_BENCH_FRAME_POOL=$pool JSP_MSV1_STAGGER=$st timeout -k 10 300 python bench.py --workload $w --steps 20 --warmup 3 --no-e2e --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], d.get('verified'), r['frac'])"
  done
 done
done
