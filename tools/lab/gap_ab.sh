#!/bin/bash
# lab: bytes left free between the frames of the batch's stream buffer (JSP_MSV1_FRAME_GAP): all-solid frames are 2^18 bytes apart without
R="${GRAFT_REPO_ROOT:-$(pwd)}"; cd $R
export JSP_BENCH_CLIPS=1
for w in ${WORKLOADS:-msvideo1_16_1080p_keyframes_solid msvideo1_16_1080p_keyframes_m1}; do
 for r in 1 2; do
  for g in 0 256 4352 20736; do
   printf "%-40s gap %6d: " $w $g
   JSP_MSV1_FRAME_GAP=$g timeout -k 10 300 python bench.py --workload $w --steps 20 --warmup 3 --no-e2e --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], d.get('verified'), r['frac'])"
  done
 done
done
