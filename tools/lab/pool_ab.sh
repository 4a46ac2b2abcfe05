#!/bin/bash
# lab: destination frames as one torch tensor each (torch), back to back in one torch allocation (1), from the product's placed frame pool (probed); torch is the default
R="${GRAFT_REPO_ROOT:-$(pwd)}"; cd $R
export JSP_BENCH_CLIPS=1
for w in ${WORKLOADS:-msvideo1_16_1080p_keyframes_m1 msvideo1_16_1080p_keyframes_solid screenpressor_v4_1080p_iframes}; do
 for r in $(seq 1 ${ROUNDS:-2}); do
  for sc in torch 1 probed; do
   printf "%-40s %-10s " $w "$sc"
   JSP_BENCH_FRAME_POOL=$sc timeout -k 10 300 python bench.py --workload $w --steps 20 --warmup 3 --no-e2e --no-cpu-baseline 2>/tmp/pool_ab.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], d.get('verified'), r['frac'], r['measured_ceiling']['value'])"
   grep -h "frame pool" /tmp/pool_ab.err || true
  done
 done
done
