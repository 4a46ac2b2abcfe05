#!/bin/bash
# lab: VALU/SALU/LDS instruction counts of msv1_fused_kernel<16,0> when it stops after phase k (k = 0..5), and whole
R="${GRAFT_REPO_ROOT:-$(pwd)}"; O="$R/gpurun_out"; export TMPDIR=/tmp
for k in ${PHASES:-0 1 2 3 4 5 9}; do
  rm -rf /tmp/alt && mkdir /tmp/alt && cp -r $R/jsplayer_amd $R/include $R/bench.py $R/tests $R/oracle $R/profiles $R/__graft_entry__.py /tmp/alt/ 2>/dev/null
  (cd /tmp/alt/jsplayer_amd/csrc && touch msv1_parse_kernels.hip && make HOOKS="-I$R/tools/lab/hooks_clocks -DJSP_FUSED_STOP=$k" > /tmp/alt/make.log 2>&1 || tail -5 /tmp/alt/make.log)
  cd /tmp
  rm -rf /tmp/pc
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/pc -- python3 /tmp/alt/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-verify > /dev/null 2> /tmp/pc.err
  python3 - $k <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob("/tmp/pc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "msv1_fused_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
w = sum(acc["SQ_WAVES"]) / max(len(acc["SQ_WAVES"]), 1)
print("stop", sys.argv[1], {k: round(sum(v) / len(v) / w, 1) for k, v in sorted(acc.items()) if k != "SQ_WAVES"}, "waves", w)
PY
done
