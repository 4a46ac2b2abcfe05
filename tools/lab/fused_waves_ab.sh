#!/bin/bash
# lab: the fused MSVideo1 kernel allowed 5 workgroups per CU (-DJSP_FUSED_WAVES=5: 96 VGPRs, 15 spilled) against the tree's 4 (111 VGPRs), same call, alternating;
# workloads: the default (M1, 3 clips), all-8-colour, 8-bit.  step ms | frac | verified
R="${GRAFT_REPO_ROOT:-$(pwd)}"
rm -rf /tmp/alt5 && mkdir /tmp/alt5 && cp -r $R/jsplayer_amd $R/include $R/bench.py $R/tests $R/oracle $R/profiles $R/__graft_entry__.py /tmp/alt5/ 2>/dev/null
(cd /tmp/alt5/jsplayer_amd/csrc && rm -f msv1_parse_kernels.o && make CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-function --offload-arch=gfx950 -DJSP_FUSED_WAVES=5" > /tmp/alt5/make.log 2>&1 || tail -5 /tmp/alt5/make.log)
one() { (cd $1 && python bench.py --workload $2 --steps 30 --warmup 5 --no-e2e --no-cpu-baseline --no-also 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['verified'])"); }
for i in 1 2; do
  for w in msvideo1_16_1080p_keyframes_m1 msvideo1_16_1080p_keyframes_eight msvideo1_8_1080p_keyframes_m1 msvideo1_16_1080p_keyframes_solid; do
    echo -n "$w 4 per CU: "; one $R $w
    echo -n "$w 5 per CU: "; one /tmp/alt5 $w
  done
done
