set -e
mkdir -p gpurun_out
B="python bench.py --workload screenpressor_v4_1080p_iframes --steps 20 --warmup 3 --no-cpu-baseline"
echo "== default" >> gpurun_out/sweep.log; timeout -k 10 200 $B >> gpurun_out/sweep.log 2>&1
echo "== ppl8" >> gpurun_out/sweep.log; JSP_SP_IFRAME_PPL=8 timeout -k 10 200 $B >> gpurun_out/sweep.log 2>&1
echo "== lds72" >> gpurun_out/sweep.log; JSP_SP_IFRAME_LDS_KB=72 timeout -k 10 200 $B >> gpurun_out/sweep.log 2>&1
echo "== lds47" >> gpurun_out/sweep.log; JSP_SP_IFRAME_LDS_KB=47 timeout -k 10 200 $B >> gpurun_out/sweep.log 2>&1
echo "== ppl8 lds72" >> gpurun_out/sweep.log; JSP_SP_IFRAME_PPL=8 JSP_SP_IFRAME_LDS_KB=72 timeout -k 10 200 $B >> gpurun_out/sweep.log 2>&1
echo "== ppl8 lds40" >> gpurun_out/sweep.log; JSP_SP_IFRAME_PPL=8 JSP_SP_IFRAME_LDS_KB=40 timeout -k 10 200 $B >> gpurun_out/sweep.log 2>&1
echo "== band135" >> gpurun_out/sweep.log; JSP_SP_IFRAME_BAND_ROWS=135 timeout -k 10 200 $B >> gpurun_out/sweep.log 2>&1
