#!/bin/bash
# lab: default workload step time with the library's kernels built with other compiler flags
R="${GRAFT_REPO_ROOT:-$(pwd)}"
echo -n "tree            "; (cd $R && python bench.py --steps 30 --warmup 5 --no-e2e --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['verified'])")
for F in "-O3 -fno-unroll-loops" "-O2 -fno-unroll-loops" "-Os -fno-unroll-loops" "-O3 -fno-unroll-loops -mllvm -amdgpu-schedule-metric-bias=0" "-O3 -fno-unroll-loops -fno-vectorize -fno-slp-vectorize"; do
  rm -rf /tmp/alt && mkdir /tmp/alt && cp -r $R/jsplayer_amd $R/include $R/bench.py $R/tests $R/oracle $R/profiles $R/__graft_entry__.py /tmp/alt/ 2>/dev/null
  (cd /tmp/alt/jsplayer_amd/csrc && touch msv1_parse_kernels.hip && make CXXFLAGS="$F -std=c++17 -fPIC -Wall -Wno-unused-function --offload-arch=gfx950" > /tmp/alt/make.log 2>&1 || tail -3 /tmp/alt/make.log)
  echo -n "$F   "; (cd /tmp/alt && python bench.py --steps 30 --warmup 5 --no-e2e --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['verified'])")
done
echo -n "tree            "; (cd $R && python bench.py --steps 30 --warmup 5 --no-e2e --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['verified'])")
