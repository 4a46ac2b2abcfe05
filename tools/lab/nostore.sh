#!/bin/bash
# lab: step time of the fused kernel with and without its frame stores (-DJSP_FUSED_NOSTORE)
R="${GRAFT_REPO_ROOT:-$(pwd)}"
run() {  # $1 = label, $2 = kernel source, $3 = extra flags
  rm -rf /tmp/alt && mkdir /tmp/alt && cp -r $R/jsplayer_amd $R/include $R/bench.py $R/tests $R/oracle $R/profiles $R/__graft_entry__.py /tmp/alt/ 2>/dev/null
  cp $2 /tmp/alt/jsplayer_amd/csrc/msv1_parse_kernels.hip
  (cd /tmp/alt/jsplayer_amd/csrc && make CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-function --offload-arch=gfx950 $3" > /tmp/alt/make.log 2>&1 || tail -5 /tmp/alt/make.log)
  for i in 1 2; do echo -n "$1 "; (cd /tmp/alt && python bench.py --steps 30 --warmup 5 --no-e2e --no-cpu-baseline --no-verify | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"); done
}
run "current        " $R/jsplayer_amd/csrc/msv1_parse_kernels.hip ""
run "current nostore" $R/jsplayer_amd/csrc/msv1_parse_kernels.hip "-DJSP_FUSED_NOSTORE"
