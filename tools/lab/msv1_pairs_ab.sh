#!/bin/bash
# MSVideo1 end to end (examples/jsp_play: AVI bytes in pinned memory -> frames in HBM, one frame per call, 8 in flight per stream, files prefetched in 32 MB ranges) with
# one-launch frames going out K to a launch (option msv1_async_pairs; lab: JSP_MSV1_FRAMES_PER_LAUNCH=K — every frame's parse behind the painting of the frames before it,
# one launch gap per K frames), K = 1 (one by one) .. 4, alternating.  one stream Mpx/s (GB/s uploaded) | 16 streams Mpx/s (GB/s uploaded)
R="${GRAFT_REPO_ROOT:-$(pwd)}"
one() { (cd $R && JSP_MSV1_FRAMES_PER_LAUNCH=$1 python bench.py --workload $2 --steps 3 --warmup 1 --no-cpu-baseline --no-also 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); e=d['e2e']; a=e['all_threads']
print(e['value'], round(e['uploaded_bytes_per_s']/1e9,1), '|', a['value'], round(a['uploaded_bytes_per_s']/1e9,1))"); }
for w in msvideo1_16_1080p_keyframes_m1 msvideo1_16_1080p_inter70 msvideo1_8_1080p_keyframes_m1; do
  for k in 1 2 3 4; do
    echo -n "$w $k per launch: "; one $k $w
  done
done
