#!/bin/bash
# lab: the tree as it is against a snapshot of an earlier commit built under .lab_prev/ (git archive <commit> jsplayer_amd include bench.py
# tests oracle profiles __graft_entry__.py | tar -x -C .lab_prev; make -C .lab_prev/jsplayer_amd/csrc), alternately, in one GPU-box call.
#   WORKLOADS="a b" STEPS=20 ROUNDS=3 tools/lab/ab_prev.sh
R="${GRAFT_REPO_ROOT:-$(pwd)}"; O="$R/gpurun_out"; mkdir -p "$O"
STEPS="${STEPS:-20}"; ROUNDS="${ROUNDS:-3}"
for w in ${WORKLOADS:-screenpressor_v4_1080p_iframes}; do
  for r in $(seq 1 $ROUNDS); do
    for side in prev new; do
      d="$R"; extra=""; [ $side = prev ] && { d="$R/.lab_prev"; extra="--no-verify"; }
      printf "%-5s %-36s " $side $w
      (cd $d && timeout -k 10 300 python bench.py --workload $w --steps $STEPS --warmup 3 --no-e2e --no-cpu-baseline $extra 2>> "$O/ab_prev.err" | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], d.get('verified'), r['frac'], r.get('kernel'), r.get('avg_us'))") || exit 1
    done
  done
done
