#!/bin/bash
# examples/jsp_play on 16 / 32 / 48 MSVideo1 player streams (M1 key frames, 8 frames in flight each, files prefetched) with 1 / 2 / 4 frames per launch
# (lab: JSP_MSV1_FRAMES_PER_LAUNCH): does grouping hurt when many launches wait for slots?
R="${GRAFT_REPO_ROOT:-$(pwd)}"
cd $R
python tools/write_workload_avi.py msvideo1_16_1080p_keyframes_m1 64 /tmp/m1.avi
for n in 4 16 32 48; do
  for k in 1 2 4; do
    echo -n "$n streams, $k per launch: "
    JSP_MSV1_FRAMES_PER_LAUNCH=$k examples/jsp_play /tmp/m1.avi --pipelined --quiet --depth 8 --streams $n --seconds 1.5 | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['mpixels_per_s'], 'Mpx/s', round(d['uploaded_bytes_per_s']/1e9,1), 'GB/s | reruns', d['async_reruns'])"
  done
done
