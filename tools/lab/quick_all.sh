#!/bin/bash
# lab: every bench workload once, resident-input numbers only (no e2e leg, no CPU baseline)
R="${GRAFT_REPO_ROOT:-$(pwd)}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
for w in ${WORKLOADS:-msvideo1_16_1080p_keyframes_m1 msvideo1_16_1080p_keyframes_m1_hostdesc msvideo1_8_1080p_keyframes_m1 msvideo1_16_1080p_keyframes_solid msvideo1_16_1080p_keyframes_eight msvideo1_16_1080p_inter70 screenpressor_v4_1080p_iframes screenpressor_v2_1080p_iframes screenpressor_v4_1080p_pclip300}; do
  timeout -k 10 600 python bench.py --workload $w --steps 20 --warmup 3 --no-e2e --no-cpu-baseline 2>> "$O/quick_all.err" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-46s %8.1f us  frac %.3f  %9.0f Mpx/s  verified %s  %s' % (d['config']['workload'], r['step_us'], r['frac'], d['value'], d['verified'], r['kernel']))"
done
