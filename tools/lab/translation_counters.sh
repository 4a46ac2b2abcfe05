#!/bin/bash
# Address-translation counters of the frame-placement lab (tools/front_lab.bin <frames> <pools>: several pools of one process, fast and slow ones among them if the
# session has both): does a slow pool miss the translation caches more?  Three rocprofv3 --pmc passes (counters only) over the same program.
# -> gpurun_out/<TAG>_utcl_*.{out,csv}
set -o pipefail
R="${GRAFT_REPO_ROOT:-$(pwd)}"; O="$R/gpurun_out"; T="${TAG:-tr}"; export TMPDIR=/tmp
cd /tmp
"$R/tools/front_lab.bin" 512 6 > "$O/${T}_front_lab_plain.txt" 2>&1 || { tail -5 "$O/${T}_front_lab_plain.txt"; exit 1; }
grep "^pool" "$O/${T}_front_lab_plain.txt" | head -12
i=0
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_THRASHING_STALL_sum" "GRBM_UTCL2_BUSY TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d "$O/${T}_utcl_$i" -- "$R/tools/front_lab.bin" 512 6 > "$O/${T}_utcl_$i.out" 2> "$O/${T}_utcl_$i.err" || { tail -5 "$O/${T}_utcl_$i.err"; }
  grep "^pool" "$O/${T}_utcl_$i.out" | head -6
done
echo done
