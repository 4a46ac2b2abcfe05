#!/bin/bash
# MSVideo1 end to end (examples/jsp_play over the C ABI: AVI bytes in pinned memory -> frames in HBM, one frame per call) with the file going up
# in ranges of N MB ahead of the frames (jsp_prefetch, jsp_play --prefetch N) against a copy / a bus read per frame (N = 0).
# one stream Mpx/s (GB/s uploaded) | 16 streams Mpx/s (GB/s uploaded) | h2d ceiling GB/s
R="${GRAFT_REPO_ROOT:-$(pwd)}"
one() { (cd $R && JSP_BENCH_PREFETCH_MB=$1 python bench.py --workload ${2:-msvideo1_16_1080p_keyframes_m1} --steps 3 --warmup 1 --no-cpu-baseline --no-also 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); e=d['e2e']; a=e['all_threads']
print(e['value'], round(e['uploaded_bytes_per_s']/1e9,1), '|', a['value'], round(a['uploaded_bytes_per_s']/1e9,1), '|', e['h2d_ceiling_GBs']['value'])"); }
for i in 1 2; do
  for n in 0 4 16 32 64; do
    echo -n "keyframes_m1, ranges of $n MB: "; one $n
  done
done
for n in 0 16; do echo -n "inter70, ranges of $n MB: "; one $n msvideo1_16_1080p_inter70; done
for n in 0 16; do echo -n "8-bit, ranges of $n MB: "; one $n msvideo1_8_1080p_keyframes_m1; done
