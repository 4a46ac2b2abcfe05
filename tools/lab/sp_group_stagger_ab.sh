#!/bin/bash
# The inter-frame group kernel with its workgroups starting in eight classes N x ~0.5 us apart (-DJSP_SP_LAB_STAGGER=N: the launch's write fronts then stand in
# different frames instead of all in the same one) against the tree, alternating; 2 x 299 frames at 1080p, digests verified.  step ms | frac | verified | pool probe
R="${GRAFT_REPO_ROOT:-$(pwd)}"
alt() { rm -rf $1 && mkdir $1 && cp -r $R/jsplayer_amd $R/include $R/bench.py $R/tests $R/oracle $R/profiles $R/__graft_entry__.py $1/ 2>/dev/null
  (cd $1/jsplayer_amd/csrc && rm -f sp_kernels.o && make CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-function --offload-arch=gfx950 $2" > $1/make.log 2>&1 || tail -5 $1/make.log); }
alt /tmp/alt_st2 -DJSP_SP_LAB_STAGGER=2
alt /tmp/alt_st4 -DJSP_SP_LAB_STAGGER=4
alt /tmp/alt_st16 -DJSP_SP_LAB_STAGGER=16
one() { (cd $1 && python bench.py --workload screenpressor_v4_1080p_pclip300 --steps 20 --warmup 5 --no-e2e --no-cpu-baseline --no-also 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['verified'], d['config']['destination_frames']['probe_GBs'])"); }
for i in 1 2; do
  echo -n "tree:        "; one $R
  echo -n "stagger 2:   "; one /tmp/alt_st2
  echo -n "stagger 4:   "; one /tmp/alt_st4
  echo -n "stagger 16:  "; one /tmp/alt_st16
done
