#!/bin/bash
# Same-call A/B of the ScreenPressor key-frame kernel with a look-ahead touch (the window after the one being fetched pulled towards the caches by one LDS-DMA
# load per window, issued as asm so that the compiler's vmcnt bookkeeping does not see it): builds with -DJSP_SP_LAB_TOUCH=8u (a word of every 64-byte line,
# 4 KB) and =4u (2 KB) against the tree, alternating; v4 and v2 key frames, digests verified.  step ms | frac | verified | pool probe
R="${GRAFT_REPO_ROOT:-$(pwd)}"
alt() {   # alt <dir> <flag>
  rm -rf $1 && mkdir $1 && cp -r $R/jsplayer_amd $R/include $R/bench.py $R/tests $R/oracle $R/profiles $R/__graft_entry__.py $1/ 2>/dev/null
  (cd $1/jsplayer_amd/csrc && rm -f sp_kernels.o && make CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-function --offload-arch=gfx950 $2" > $1/make.log 2>&1 || tail -5 $1/make.log)
}
alt /tmp/alt_touch8 -DJSP_SP_LAB_TOUCH=8u
alt /tmp/alt_touch4 -DJSP_SP_LAB_TOUCH=4u
one() { (cd $1 && python bench.py --workload $2 --steps 30 --warmup 5 --no-e2e --no-cpu-baseline --no-also 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['verified'], d['config']['destination_frames']['probe_GBs'])"); }
for i in 1 2 3; do
  for w in screenpressor_v4_1080p_iframes screenpressor_v2_1080p_iframes; do
    echo -n "$w tree (no touch): "; one $R $w
    echo -n "$w touch 4 KB:      "; one /tmp/alt_touch8 $w
    echo -n "$w touch 2 KB:      "; one /tmp/alt_touch4 $w
  done
done
