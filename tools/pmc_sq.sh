#!/bin/bash
# Shader-core counters of one bench workload (three rocprofv3 --pmc passes, nothing else traced), summed per kernel.
# Usage: tools/pmc_sq.sh <workload> <kernel substring>   -> gpurun_out/pmc_sq_<workload>.txt
set -eo pipefail
R="${GRAFT_REPO_ROOT:-$(pwd)}"; O="$R/gpurun_out"; mkdir -p "$O"
W="$1"; K="$2"; export TMPDIR=/tmp
cd /tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d "$O/pmc_sq_$i" -- python3 "$R/bench.py" --workload "$W" --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-verify > /dev/null 2> "$O/pmc_sq_$i.err"
done
cd "$R"
python3 - "$O" "$K" "$W" <<'PY'
import csv, glob, os, sys, collections
O, K, W = sys.argv[1:4]
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(O, "pmc_sq_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if K in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(O, f"pmc_sq_{W}.txt"), "w") as out:
    out.write(f"workload {W}, kernel *{K}*, mean per dispatch over {max(len(v) for v in acc.values()) if acc else 0} dispatches\n")
    for k in sorted(acc):
        out.write(f"{k:34s} {sum(acc[k]) / len(acc[k]):16.0f}\n")
print(open(os.path.join(O, f"pmc_sq_{W}.txt")).read())
PY
rm -rf "$O"/pmc_sq_[0-9]*
