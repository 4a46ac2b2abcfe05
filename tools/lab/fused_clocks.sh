#!/bin/bash
# Per-phase cycle sums of msv1_fused_kernel (a build whose msv1_fused_hooks.h is tools/lab/hooks_clocks': thread 0 of every tile adds the cycles of each phase to
# counters behind the tile tables; printed per tile and launch at every sync) for the M1 mix, the all-eight-colour mix, the 8-bit
# and the solid workloads.  The clocks cost time themselves: the step times beside them are not the product's.
R="${GRAFT_REPO_ROOT:-$(pwd)}"
rm -rf /tmp/alt_clocks && mkdir /tmp/alt_clocks && cp -r $R/jsplayer_amd $R/include $R/bench.py $R/tests $R/oracle $R/profiles $R/__graft_entry__.py /tmp/alt_clocks/ 2>/dev/null
(cd /tmp/alt_clocks/jsplayer_amd/csrc && rm -f msv1_parse_kernels.o msv1_codec.o && make HOOKS="-I$R/tools/lab/hooks_clocks" > /tmp/alt_clocks/make.log 2>&1 || tail -5 /tmp/alt_clocks/make.log)
for w in ${WORKLOADS:-msvideo1_16_1080p_keyframes_m1 msvideo1_16_1080p_keyframes_eight msvideo1_8_1080p_keyframes_m1 msvideo1_16_1080p_keyframes_solid}; do
  echo "== $w"
  (cd /tmp/alt_clocks && python bench.py --workload $w --steps 10 --warmup 2 --no-e2e --no-cpu-baseline --no-also 2>/tmp/alt_clocks/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'verified', d['verified'])"; grep "fused clocks" /tmp/alt_clocks/err.txt | tail -2)
done
