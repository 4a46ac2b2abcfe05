#!/bin/bash
# lab: shader-core counters (tools/pmc_sq.sh) of one workload's kernel for the snapshot under .lab_prev/ and for the tree, in one GPU-box call.
#   tools/lab/sq_ab.sh <workload> <kernel substring> <tag>
R="${GRAFT_REPO_ROOT:-$(pwd)}"; O="$R/gpurun_out"; mkdir -p "$O" "$R/.lab_prev/gpurun_out"
W="$1"; K="$2"; T="${3:-ab}"
GRAFT_REPO_ROOT="$R/.lab_prev" "$R/tools/pmc_sq.sh" "$W" "$K" > /dev/null && cp "$R/.lab_prev/gpurun_out/pmc_sq_$W.txt" "$O/${T}_sq_prev_$W.txt"
GRAFT_REPO_ROOT="$R" "$R/tools/pmc_sq.sh" "$W" "$K" > /dev/null && cp "$O/pmc_sq_$W.txt" "$O/${T}_sq_new_$W.txt"
paste "$O/${T}_sq_prev_$W.txt" "$O/${T}_sq_new_$W.txt" | awk 'NR==1{print; next}{printf "%-34s %16s %16s  %6.3f\n", $1, $2, $4, ($2>0?$4/$2:0)}' | tee "$O/${T}_sq_ab_$W.txt"
