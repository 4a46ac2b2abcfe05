#!/bin/bash
# lab: shader-core counters of the ScreenPressor key-frame kernel, whole against the build without its row stores (same call): where do the
# waves of the whole kernel spend the time the arithmetic-only build does not?
R="${GRAFT_REPO_ROOT:-$(pwd)}"; O="$R/gpurun_out"; export TMPDIR=/tmp
rm -rf /tmp/alt_nostore && mkdir /tmp/alt_nostore && cp -r $R/jsplayer_amd $R/include $R/bench.py $R/tests $R/oracle $R/profiles $R/__graft_entry__.py /tmp/alt_nostore/ 2>/dev/null
(cd /tmp/alt_nostore/jsplayer_amd/csrc && rm -f sp_kernels.o && make CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-function --offload-arch=gfx950 -DJSP_SP_LAB_NOSTORE" > /tmp/alt_nostore/make.log 2>&1 || tail -5 /tmp/alt_nostore/make.log)
cd /tmp
for which in whole nostore; do
  D=$R; [ $which = nostore ] && D=/tmp/alt_nostore
  i=0
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
             "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d "$O/sq_${which}_$i" -- python3 "$D/bench.py" --workload screenpressor_v4_1080p_iframes --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-verify > /dev/null 2> "$O/sq_${which}_$i.err"
  done
done
cd "$R"
python3 - "$O" <<'PY'
import csv, glob, os, sys, collections
O = sys.argv[1]
out = open(os.path.join(O, "sp_tile_sq_whole_vs_nostore.txt"), "w")
for which in ("whole", "nostore"):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(O, f"sq_{which}_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "sp_iframe_tile_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    out.write(f"-- {which}: sp_iframe_tile_kernel, mean per dispatch\n")
    for k in sorted(acc):
        out.write(f"{k:34s} {sum(acc[k]) / len(acc[k]):16.0f}\n")
out.close()
print(open(os.path.join(O, "sp_tile_sq_whole_vs_nostore.txt")).read())
PY
rm -rf "$O"/sq_whole_* "$O"/sq_nostore_*
