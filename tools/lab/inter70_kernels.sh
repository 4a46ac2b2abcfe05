#!/bin/bash
# lab: kernel durations of the inter-frame workload with the launches in line (no parse ahead), compact tables on / off: rocprofv3 kernel stats of each
R="${GRAFT_REPO_ROOT:-$(pwd)}"; O="$R/gpurun_out"; T="${TAG:-lab}"; export TMPDIR=/tmp
cd /tmp
for c in 1 0; do
  JSP_MSV1_PARSE_AHEAD=0 JSP_MSV1_COMPACT_TABLES=$c timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_c$c" -o p -- python3 "$R/bench.py" --workload msvideo1_16_1080p_inter70 --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-also --no-verify > "$O/${T}_inter70_compact${c}_bench.json" 2> "$O/${T}_inter70_compact${c}.err" || { tail -5 "$O/${T}_inter70_compact${c}.err"; exit 1; }
  find "$O/prof_c$c" -name "*kernel_stats.csv" -exec cp {} "$O/${T}_inter70_compact${c}_kernel_stats.csv" \;
  rm -rf "$O/prof_c$c"
  echo "compact tables $c:"; head -4 "$O/${T}_inter70_compact${c}_kernel_stats.csv" | cut -c1-160
done
