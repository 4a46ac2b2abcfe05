#!/bin/bash
# Tuning sweep of the ScreenPressor key-frame kernels (64 x 1080p key frames per launch): kernel variant,
# LDS budget (= workgroups per CU), band height.  Appends one bench line per setting to gpurun_out/sweep.log.
set -e
mkdir -p gpurun_out
B="python bench.py --workload screenpressor_v4_1080p_iframes --steps 20 --warmup 3 --no-cpu-baseline"
run() { echo "== $1" >> gpurun_out/sweep.log; shift; env "$@" timeout -k 10 200 $B >> gpurun_out/sweep.log 2>&1; }
run "default (reg kernel, 40 KB, auto bands)" X=1
run "reg lds32" JSP_SP_IFRAME_LDS_KB=32
run "reg lds52" JSP_SP_IFRAME_LDS_KB=52
run "reg band24" JSP_SP_IFRAME_BAND_ROWS=24
run "reg band68" JSP_SP_IFRAME_BAND_ROWS=68
run "reg band135" JSP_SP_IFRAME_BAND_ROWS=135
run "reg one band per frame" JSP_SP_IFRAME_BAND_ROWS=0
run "rows kernel (row above in LDS), 4 px per lane" JSP_SP_IFRAME_KERNEL=rows
run "rows kernel, 8 px per lane" JSP_SP_IFRAME_KERNEL=rows JSP_SP_IFRAME_PPL=8
run "search kernel" JSP_SP_IFRAME_KERNEL=search
