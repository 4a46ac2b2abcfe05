#!/bin/bash
# lab: several builds of the library, each with its own msv1_parse_kernels.hip / extra compiler flags, timed alternately in one call.
#   tools/lab/variants.sh "label|source.hip|flags" ...      (source relative to the repo root; empty = the tree's file)
#   WORKLOAD=... STEPS=... ROUNDS=... as environment.  Builds run side by side; stderr of the bench (lab clocks) is kept per label.
R="${GRAFT_REPO_ROOT:-$(pwd)}"; O="$R/gpurun_out"; mkdir -p "$O"
W="${WORKLOAD:-msvideo1_16_1080p_keyframes_m1}"; STEPS="${STEPS:-30}"; ROUNDS="${ROUNDS:-3}"
n=0
for spec in "$@"; do
  IFS='|' read -r label src flags <<< "$spec"
  d=/tmp/alt_$n; rm -rf $d; mkdir $d
  cp -r $R/jsplayer_amd $R/include $R/bench.py $R/tests $R/oracle $R/profiles $R/__graft_entry__.py $d/ 2>/dev/null
  [ -n "$src" ] && cp "$R/$src" $d/jsplayer_amd/csrc/msv1_parse_kernels.hip
  (cd $d/jsplayer_amd/csrc && touch msv1_parse_kernels.hip && make -j4 CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-function --offload-arch=gfx950 $flags" > $d/make.log 2>&1 || tail -5 $d/make.log) &
  labels[$n]="$label"; n=$((n+1))
done
wait
for r in $(seq 1 $ROUNDS); do
  for k in $(seq 0 $((n-1))); do
    printf "%-22s " "${labels[$k]}"
    (cd /tmp/alt_$k && python bench.py --workload $W --steps $STEPS --warmup 5 --no-e2e --no-cpu-baseline ${EXTRA:-} 2> "$O/lab_${labels[$k]// /_}.err" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['verified'], d['roofline']['frac'])")
    grep -h "fused clocks" "$O/lab_${labels[$k]// /_}.err" | tail -1
  done
done
