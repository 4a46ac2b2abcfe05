#!/bin/bash
# lab: A/B of the tree's library against one built with another sp_kernels.hip ($1), workload $2, alternating runs
R="${GRAFT_REPO_ROOT:-$(pwd)}"
rm -rf /tmp/alt && mkdir /tmp/alt && cp -r $R/jsplayer_amd $R/include $R/bench.py $R/tests $R/oracle $R/profiles $R/__graft_entry__.py /tmp/alt/ 2>/dev/null
cp $1 /tmp/alt/jsplayer_amd/csrc/sp_kernels.hip
(cd /tmp/alt/jsplayer_amd/csrc && make > /tmp/alt/make.log 2>&1 || tail -5 /tmp/alt/make.log)
for i in 1 2 3; do
  echo -n "tree "; (cd $R && python bench.py --workload $2 --steps 30 --warmup 5 --no-e2e --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['verified'], d['roofline']['frac'])")
  echo -n "alt  "; (cd /tmp/alt && python bench.py --workload $2 --steps 30 --warmup 5 --no-e2e --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['verified'], d['roofline']['frac'])")
done
