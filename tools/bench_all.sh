#!/bin/bash
# Every bench workload once (N = 1), one JSON line each -> gpurun_out/bench_all.jsonl
set -eo pipefail
mkdir -p gpurun_out; : > gpurun_out/bench_all.jsonl
for w in msvideo1_16_1080p_keyframes_m1 msvideo1_8_1080p_keyframes_m1 msvideo1_16_1080p_keyframes_solid msvideo1_16_1080p_keyframes_eight \
         msvideo1_16_1080p_keyframes_m1_gpuparse msvideo1_16_1080p_inter70 screenpressor_v4_1080p_iframes screenpressor_v4_1080p_iframes_x8 \
         screenpressor_v2_1080p_iframes screenpressor_v4_1080p_pclip300; do
  steps=100; case $w in screenpressor*) steps=20;; esac
  timeout -k 10 500 python bench.py --workload $w --steps $steps --warmup 5 >> gpurun_out/bench_all.jsonl 2>> gpurun_out/bench_all.err
done
