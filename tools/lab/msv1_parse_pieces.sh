#!/bin/bash
# The inter-frame batch (msvideo1_16_1080p_inter70) with the replay's table-writing parse in 1 / 2 / 4 / 8 / 16 pieces on a second
# stream beside the temporal launches (option msv1_parse_pieces), alternating, digests verified.  step ms | frac | verified | look-back fallbacks | pool probe
R="${GRAFT_REPO_ROOT:-$(pwd)}"
one() { (cd $R && JSP_BENCH_OPTIONS="msv1_parse_pieces=$1" python bench.py --workload ${2:-msvideo1_16_1080p_inter70} --steps 30 --warmup 5 --no-e2e --no-cpu-baseline --no-also 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['verified'], d.get('lookback_fallbacks'), d['config']['destination_frames']['probe_GBs'])"); }
for i in 1 2; do
  for n in 1 2 4 8 16; do
    echo -n "inter70, $n piece(s): "; one $n
  done
done
