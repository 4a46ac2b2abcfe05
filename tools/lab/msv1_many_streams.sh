#!/bin/bash
# How many MSVideo1 player streams does one GPU take?  examples/jsp_play on 1 / 4 / 16 / 32 / 48 streams (host threads; the box has 16 CPUs), M1 key frames and the
# inter-frame clip, 8 frames in flight per stream, files prefetched, frames four to a launch: rate, and how many frames the host had to re-run (a one-launch
# frame's tiles all have to be resident together for its verdict: with enough launches in flight they would wait for each other's slots until a time-out).
R="${GRAFT_REPO_ROOT:-$(pwd)}"
cd $R
python tools/write_workload_avi.py msvideo1_16_1080p_keyframes_m1 64 /tmp/m1.avi
python tools/write_workload_avi.py msvideo1_16_1080p_inter70 96 /tmp/i70.avi
for f in /tmp/m1.avi /tmp/i70.avi; do
  for n in 1 4 16 32 48; do
    echo -n "$f $n streams: "
    examples/jsp_play $f --pipelined --quiet --depth 8 --streams $n --seconds 1.5 | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['mpixels_per_s'], 'Mpx/s', round(d['uploaded_bytes_per_s']/1e9,1), 'GB/s | frames', d['frames'], 'reruns', d['async_reruns'], 'shared a launch', d['paired_frames'])"
  done
done
