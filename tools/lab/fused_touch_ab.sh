#!/bin/bash
# msv1_fused_kernel (batch form) with every tile touching the stream bytes of the tile N launches ahead (-DJSP_FUSED_LAB_TOUCH=N: one LDS-DMA load per thread
# into a sink, issued as asm behind the tile's own loads) against the tree, alternating; digests verified.  step ms | frac | verified
R="${GRAFT_REPO_ROOT:-$(pwd)}"
alt() { rm -rf $1 && mkdir $1 && cp -r $R/jsplayer_amd $R/include $R/bench.py $R/tests $R/oracle $R/profiles $R/__graft_entry__.py $1/ 2>/dev/null
  (cd $1/jsplayer_amd/csrc && rm -f msv1_parse_kernels.o && make CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-function --offload-arch=gfx950 $2" > $1/make.log 2>&1 || tail -5 $1/make.log); }
alt /tmp/alt_ft256 -DJSP_FUSED_LAB_TOUCH=256
alt /tmp/alt_ft1024 -DJSP_FUSED_LAB_TOUCH=1024
one() { (cd $1 && python bench.py --workload $2 --steps 30 --warmup 5 --no-e2e --no-cpu-baseline --no-also 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['verified'])"); }
for i in 1 2; do
  for w in msvideo1_16_1080p_keyframes_eight msvideo1_8_1080p_keyframes_m1 msvideo1_16_1080p_keyframes_m1 msvideo1_16_1080p_keyframes_solid; do
    echo -n "$w tree:            "; one $R $w
    echo -n "$w touch 256 ahead: "; one /tmp/alt_ft256 $w
    echo -n "$w touch 1024 ahead:"; one /tmp/alt_ft1024 $w
  done
done
