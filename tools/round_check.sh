#!/bin/bash
# Usage: tools/round_check.sh [tests] [smoke] [bench] [all-workloads] [profile:<workload>] ...
# One GPU-box pass; every step is optional and lands in gpurun_out/<tag>_*.  TAG=<prefix> names the files.
# Steps are chained: the first failure stops the pass (no GPU step runs after a failed one).
set -eo pipefail
R="${GRAFT_REPO_ROOT:-$(pwd)}"
O="$R/gpurun_out"; mkdir -p "$O"
T="${TAG:-run}"
export TMPDIR=/tmp
cd "$R"
for step in "$@"; do
  case "$step" in
    tests)
      timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=12 > "$O/${T}_gpu_tests.log" 2>&1 || { tail -30 "$O/${T}_gpu_tests.log"; exit 1; }
      tail -2 "$O/${T}_gpu_tests.log" ;;
    newtests)
      timeout -k 10 900 python -m pytest tests/test_bench_workloads_gpu.py -m gpu -x -q > "$O/${T}_gpu_newtests.log" 2>&1 || { tail -30 "$O/${T}_gpu_newtests.log"; exit 1; }
      tail -2 "$O/${T}_gpu_newtests.log" ;;
    smoke)
      timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > "$O/${T}_smoke.log" 2>&1 || { tail -20 "$O/${T}_smoke.log"; exit 1; }
      tail -1 "$O/${T}_smoke.log" ;;
    bench)
      timeout -k 10 500 python bench.py > "$O/${T}_bench_default.json" 2> "$O/${T}_bench_default.err" || { tail -20 "$O/${T}_bench_default.err"; exit 1; }
      cat "$O/${T}_bench_default.json" ;;
    all-workloads)
      : > "$O/${T}_bench_all.jsonl"
      for w in msvideo1_16_1080p_keyframes_m1_hostdesc msvideo1_8_1080p_keyframes_m1 msvideo1_16_1080p_keyframes_solid \
               msvideo1_16_1080p_keyframes_eight msvideo1_16_1080p_inter70 screenpressor_v4_1080p_iframes \
               screenpressor_v2_1080p_iframes screenpressor_v4_1080p_pclip300; do
        timeout -k 10 600 python bench.py --workload $w --steps 20 --warmup 5 >> "$O/${T}_bench_all.jsonl" 2>> "$O/${T}_bench_all.err" || { tail -20 "$O/${T}_bench_all.err"; exit 1; }
        echo "done $w"
      done ;;
    profile:*)
      W="${step#profile:}"
      cd /tmp
      timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_$W" -o p -- python3 "$R/bench.py" --workload $W --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-also > "$O/${T}_${W}_bench_under_rocprof.json" 2> "$O/${T}_${W}_rocprof.err" || { tail -20 "$O/${T}_${W}_rocprof.err"; exit 1; }
      find "$O/prof_$W" -name "*kernel_stats.csv" -exec cp {} "$O/${T}_${W}_kernel_stats.csv" \;
      rm -rf "$O/prof_$W"
      # the same command WITHOUT the profiler, same call: the pair is what profiles/ keeps (tools/profile_pair.py)
      timeout -k 10 400 python3 "$R/bench.py" --workload $W --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-also > "$O/${T}_${W}_bench_same_call.json" 2>> "$O/${T}_${W}_rocprof.err" || { tail -20 "$O/${T}_${W}_rocprof.err"; exit 1; }
      python3 "$R/tools/profile_pair.py" "$O/${T}_${W}_kernel_stats.csv" "$O/${T}_${W}_bench_under_rocprof.json" "$O/${T}_${W}_bench_same_call.json" "$O/${T}_${W}_pair.json" || exit 1
      timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_f_$W" -- python3 "$R/bench.py" --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-verify --no-also > /dev/null 2> "$O/${T}_${W}_pmc_f.err" || { tail -20 "$O/${T}_${W}_pmc_f.err"; exit 1; }
      timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_w_$W" -- python3 "$R/bench.py" --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-verify --no-also > /dev/null 2> "$O/${T}_${W}_pmc_w.err" || { tail -20 "$O/${T}_${W}_pmc_w.err"; exit 1; }
      cd "$R"
      python tools/pmc_summary.py "$O/pmc_f_$W" "$O/pmc_w_$W" "$O/${T}_${W}_bench_under_rocprof.json" "$O/${T}_${W}_traffic.json" || exit 1
      rm -rf "$O/pmc_f_$W" "$O/pmc_w_$W"
      head -8 "$O/${T}_${W}_kernel_stats.csv"; cat "$O/${T}_${W}_traffic.json" ;;
    *) echo "unknown step $step"; exit 2 ;;
  esac
done
