#!/bin/bash
# lab: MSVideo1 end to end through the asynchronous calls (examples/jsp_play --pipelined --quiet), the frames' bytes taken up by the copy engine on
# 1 / 2 / 3 / 4 streams in turn (JSP_MSV1_UP_STREAMS), and with the kernel reading the pinned bytes itself (one_launch); 1 and 2 player streams.
R="${GRAFT_REPO_ROOT:-$(pwd)}"; O="$R/gpurun_out"; T="${TAG:-ab}"
cd "$R"
python tools/write_workload_avi.py msvideo1_16_1080p_keyframes_m1 256 /tmp/m1.avi
python tools/write_workload_avi.py msvideo1_16_1080p_inter70 256 /tmp/inter.avi
python tools/write_workload_avi.py msvideo1_8_1080p_keyframes_m1 256 /tmp/m8.avi
: > "$O/${T}_msv1_up_streams.txt"
for round in 1 2; do
  for f in m1 inter m8; do
    for n in 1 2 3 4; do
      for s in 1 2; do
        echo -n "$f up_streams $n player streams $s depth 8: " | tee -a "$O/${T}_msv1_up_streams.txt"
        JSP_MSV1_UP_STREAMS=$n examples/jsp_play /tmp/$f.avi --pipelined --quiet --depth 8 --streams $s --seconds 1.5 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["mpixels_per_s"], "Mpx/s", round(d["uploaded_bytes_per_s"]/1e9,1), "GB/s")' | tee -a "$O/${T}_msv1_up_streams.txt"
      done
    done
    echo -n "$f kernel reads the pinned bytes itself (one_launch), 1 player stream: " | tee -a "$O/${T}_msv1_up_streams.txt"
    JSP_PLAY_MSV1_ASYNC=one_launch examples/jsp_play /tmp/$f.avi --pipelined --quiet --depth 8 --streams 1 --seconds 1.5 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["mpixels_per_s"], "Mpx/s", round(d["uploaded_bytes_per_s"]/1e9,1), "GB/s")' | tee -a "$O/${T}_msv1_up_streams.txt"
  done
done
