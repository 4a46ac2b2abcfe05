#!/bin/bash
# lab: msv1_fused_kernel with the lane-table pass fenced every 1 / 4 / 8 steps (42 / 42 / 52 VGPRs, five workgroups per CU by LDS), the same at FOUR workgroups per CU
# (LDS padded), and without fences (111 VGPRs, four workgroups per CU: the round-4 kernel), alternating, one call.
R="${GRAFT_REPO_ROOT:-$(pwd)}"
build() {  # name, extra flags
  rm -rf /tmp/alt_$1 && mkdir /tmp/alt_$1 && cp -r $R/jsplayer_amd $R/include $R/bench.py $R/tests $R/oracle $R/__graft_entry__.py /tmp/alt_$1/ 2>/dev/null; mkdir -p /tmp/alt_$1/profiles
  (cd /tmp/alt_$1/jsplayer_amd/csrc && rm -f msv1_parse_kernels.o msv1_codec.o && make HOOKS="-I$R/tools/lab/hooks_clocks -DJSP_FUSED_STOP=99 $2" > /tmp/alt_$1/make.log 2>&1 || { tail -5 /tmp/alt_$1/make.log; exit 1; })
}
for v in ${VARIANTS:-nofence f1 f4 f8 f4w4}; do case $v in f1) build f1 "-DJSP_LANE_FENCE_EVERY=1";; f4) build f4 "-DJSP_LANE_FENCE_EVERY=4";; f8) build f8 "-DJSP_LANE_FENCE_EVERY=8";; f4w4) build f4w4 "-DJSP_LANE_FENCE_EVERY=4 -DJSP_FUSED_LDS_PAD=1400";; nofence) build nofence "-DJSP_LANE_FENCE_EVERY=64";; esac; done
one() { (cd $1 && python bench.py --workload $2 --steps 30 --warmup 5 --no-e2e --no-cpu-baseline --no-also 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['verified'], d['config']['destination_frames']['probe_GBs'])"); }
for w in ${WORKLOADS:-msvideo1_16_1080p_keyframes_eight msvideo1_16_1080p_keyframes_m1 msvideo1_16_1080p_keyframes_solid msvideo1_8_1080p_keyframes_m1}; do
  for i in $(seq 1 ${ROUNDS:-2}); do for v in ${VARIANTS:-nofence f1 f4 f8 f4w4}; do echo -n "$v $w "; one /tmp/alt_$v $w; done; done
done
