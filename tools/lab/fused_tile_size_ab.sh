#!/bin/bash
# lab: msv1_fused_kernel's batch forms with 8 KiB tiles (JSP_BATCH_LS=16: twice the workgroups, half the serial work per tile) against the product's 16 KiB, alternating, one call.
R="${GRAFT_REPO_ROOT:-$(pwd)}"
rm -rf /tmp/alt_ls16 && mkdir /tmp/alt_ls16 && cp -r $R/jsplayer_amd $R/include $R/bench.py $R/tests $R/oracle $R/__graft_entry__.py /tmp/alt_ls16/ 2>/dev/null; mkdir -p /tmp/alt_ls16/profiles
(cd /tmp/alt_ls16/jsplayer_amd/csrc && rm -f msv1_parse_kernels.o msv1_codec.o && make HOOKS="-I$R/tools/lab/hooks_clocks -DJSP_BATCH_LS=16 -DJSP_FUSED_STOP=99" > /tmp/alt_ls16/make.log 2>&1 || { tail -5 /tmp/alt_ls16/make.log; exit 1; })
one() { (cd $1 && python bench.py --workload $2 --steps 30 --warmup 5 --no-e2e --no-cpu-baseline --no-also 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['verified'], d['config']['destination_frames']['probe_GBs'])"); }
for w in ${WORKLOADS:-msvideo1_16_1080p_keyframes_eight msvideo1_8_1080p_keyframes_m1 msvideo1_16_1080p_keyframes_m1}; do
  for i in 1 2; do
    echo -n "16KiB $w "; one $R $w
    echo -n " 8KiB $w "; one /tmp/alt_ls16 $w
  done
done
