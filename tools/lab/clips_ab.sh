#!/bin/bash
R="${GRAFT_REPO_ROOT:-$(pwd)}"; cd $R
for r in 1 2; do
 for c in 1 2 3; do
  printf "clips %d: " $c
  JSP_BENCH_CLIPS=$c timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-e2e --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], d.get('verified'), r['frac'], d['config']['frames_per_step'])"
 done
done
