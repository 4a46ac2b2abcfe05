#!/bin/bash
# lab: where sp_pframe_group_kernel's time goes — the tree against three variants of the kernel (tools/lab/sp_group_parts.py), alternating,
# one gpurun call.  step ms | frac by moved bytes | verified | pool probe
R="${GRAFT_REPO_ROOT:-$(pwd)}"; W="${W:-screenpressor_v4_1080p_pclip300}"
for v in nolit norec idle; do
  rm -rf /tmp/alt_$v && mkdir /tmp/alt_$v && cp -r $R/jsplayer_amd $R/include $R/bench.py $R/tests $R/oracle $R/__graft_entry__.py /tmp/alt_$v/ 2>/dev/null; mkdir -p /tmp/alt_$v/profiles
  python3 $R/tools/lab/sp_group_parts.py $R/jsplayer_amd/csrc/sp_kernels.hip $v /tmp/alt_$v/jsplayer_amd/csrc/sp_kernels.hip || exit 1
  (cd /tmp/alt_$v/jsplayer_amd/csrc && make > /tmp/alt_$v/make.log 2>&1 || { tail -5 /tmp/alt_$v/make.log; exit 1; })
done
one() { (cd $1 && python bench.py --workload $W --steps 30 --warmup 5 --no-e2e --no-cpu-baseline $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['verified'], d['config']['destination_frames']['probe_GBs'])"); }
for i in 1 2; do
  echo -n "tree   "; one $R ""
  for v in nolit norec idle; do echo -n "$v  "; one /tmp/alt_$v --no-verify; done
done
