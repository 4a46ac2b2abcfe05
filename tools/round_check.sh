#!/bin/bash
# Usage: tools/round_check.sh [all|profile-only] [workload] [kernel substring]
# One GPU-box pass: the whole GPU test suite, smoke(), the default bench line, then kernel-trace and
# PMC profiles of the ScreenPressor key-frame workload.  Everything lands in gpurun_out/.
set -eo pipefail
R="${GRAFT_REPO_ROOT:-$(pwd)}"
O="$R/gpurun_out"; mkdir -p "$O"
export TMPDIR=/tmp
cd "$R"
if [ "$1" != "profile-only" ]; then
timeout -k 10 900 python -m pytest tests -m gpu -x -q > "$O/gpu_tests.log" 2>&1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$O/smoke.log" 2>&1
timeout -k 10 400 python bench.py > "$O/bench_default.json" 2> "$O/bench_default.err"
fi
W="${2:-screenpressor_v4_1080p_iframes}"; K="${3:-sp_iframe_rows_kernel}"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_spi" -o spi -- python3 "$R/bench.py" --workload $W --steps 20 --warmup 3 --no-cpu-baseline > "$O/spi_bench_under_rocprof.json" 2> "$O/spi_rocprof.err"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_spi_fetch" -- python3 "$R/bench.py" --workload $W --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> "$O/spi_pmc_f.err"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_spi_write" -- python3 "$R/bench.py" --workload $W --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> "$O/spi_pmc_w.err"
cd "$R"
python tools/pmc_summary.py "$O/pmc_spi_fetch" "$O/pmc_spi_write" "$K" $W scratch > "$O/spi_traffic.json"
rm -rf "$O/pmc_spi_fetch" "$O/pmc_spi_write"
find "$O/prof_spi" -name "*kernel_stats.csv" -exec cp {} "$O/spi_kernel_stats.csv" \;
rm -rf "$O/prof_spi"
[ "$1" = "profile-only" ] || { tail -2 "$O/gpu_tests.log"; tail -1 "$O/smoke.log"; cat "$O/bench_default.json"; }
cat "$O/spi_traffic.json"; head -5 "$O/spi_kernel_stats.csv"
