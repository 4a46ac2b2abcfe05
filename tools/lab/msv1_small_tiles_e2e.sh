#!/bin/bash
# MSVideo1 end to end (jsp_play, files prefetched in 32 MB ranges) with 8 KiB tiles for one-frame launches of frames up to N bytes (lab: JSP_MSV1_SMALL_TILE_BYTES;
# the product's limit is 640 KB): megabyte key frames with 16 KiB against 8 KiB tiles.  one stream Mpx/s | 16 streams Mpx/s
R="${GRAFT_REPO_ROOT:-$(pwd)}"
one() { (cd $R && JSP_MSV1_SMALL_TILE_BYTES=$1 python bench.py --workload msvideo1_16_1080p_keyframes_m1 --steps 3 --warmup 1 --no-cpu-baseline --no-also 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); e=d['e2e']; a=e['all_threads']; print(e['value'], '|', a['value'])"); }
for i in 1 2; do
  echo -n "limit 640 KB (16 KiB tiles for these frames): "; one 655360
  echo -n "limit 1.1 MB (8 KiB tiles):                   "; one 1153434
done
