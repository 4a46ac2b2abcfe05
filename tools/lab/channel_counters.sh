#!/bin/bash
# Per-channel L2 / fabric counters of the frame-placement lab (tools/front_lab.bin <frames> <pools>: several pools of one process, a
# fast one and a slow one among them if the session has both): is a slow pool a channel imbalance?  Lists what rocprofv3 offers for
# the TCC block first, then three --pmc passes (nothing else traced).  -> gpurun_out/<TAG>_channel_*.txt
set -o pipefail
R="${GRAFT_REPO_ROOT:-$(pwd)}"; O="$R/gpurun_out"; T="${TAG:-ch}"; export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > "$O/${T}_rocprof_counters.txt" 2>&1 || true
grep -c . "$O/${T}_rocprof_counters.txt"
grep -o "TCC_[A-Z0-9_]*\(\[[0-9]*\]\)\?" "$O/${T}_rocprof_counters.txt" | sort -u > "$O/${T}_tcc_counter_names.txt"
wc -l "$O/${T}_tcc_counter_names.txt"
# the lab without a profiler first: which pools are fast, which slow
"$R/tools/front_lab.bin" 512 6 > "$O/${T}_front_lab_plain.txt" 2>&1 || { tail -5 "$O/${T}_front_lab_plain.txt"; exit 1; }
grep "^pool" "$O/${T}_front_lab_plain.txt" | grep "plain fill" | head -12
i=0
for set in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_64B_sum" "TCC_TAG_STALL_sum TCC_EA0_WRREQ_DRAM_sum TCC_BUSY_sum" "TCC_EA0_WR_UNCACHED_32B_sum TCC_WRITE_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d "$O/${T}_chan_$i" -- "$R/tools/front_lab.bin" 512 6 > "$O/${T}_chan_$i.out" 2> "$O/${T}_chan_$i.err" || { tail -5 "$O/${T}_chan_$i.err"; }
done
echo done
