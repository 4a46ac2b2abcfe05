#!/bin/bash
# ThreadSanitizer over the product's threaded HOST layers, without a GPU (GPU sanitizers are not available on this pool): jsp_api.cpp,
# msv1_codec.cpp, sp_codec.cpp, jsp_shard.cpp and the host stages built with g++ -fsanitize=thread against the stub HIP runtime and the
# kernel stubs under tests/tsan/, driven by tests/tsan/driver.cpp (asynchronous submit / wait out of phase, drains, prefetch ranges given up
# mid-flight, staged batches, sixteen-odd streams on as many threads, pools created and destroyed side by side).
# Usage: tools/tsan_cpu.sh [passes]     exit 0 and "tsan clean" = no report
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="${JSP_TSAN_DIR:-$(mktemp -d /tmp/jsp_tsan.XXXX)}"
C="$ROOT/jsplayer_amd/csrc"; T="$ROOT/tests/tsan"
FLAGS="-std=c++17 -O1 -g -fsanitize=thread -fPIC -pthread -I$T -I$C -I$ROOT/include"
pids=()
for f in jsp_api jsp_shard msv1_codec msv1_host sp_codec sp_entropy sp_host sp_models; do
  g++ $FLAGS -c "$C/$f.cpp" -o "$OUT/$f.o" & pids+=($!)
  if [ ${#pids[@]} -ge 4 ]; then wait "${pids[0]}"; pids=("${pids[@]:1}"); fi
done
for p in "${pids[@]}"; do wait "$p"; done
g++ $FLAGS -c "$T/hip_stub.cpp" -o "$OUT/hip_stub.o"
g++ $FLAGS -c "$T/kernel_stubs.cpp" -o "$OUT/kernel_stubs.o"
g++ $FLAGS "$T/driver.cpp" "$OUT"/*.o -o "$OUT/tsan_driver" -ldl
python3 "$T/make_clips.py" "$OUT/clips.bin" > /dev/null
cd "$OUT"
set +e
TSAN_OPTIONS="halt_on_error=0 exitcode=66 history_size=4" timeout 900 ./tsan_driver "$OUT/clips.bin" "${1:-3}" > "$OUT/tsan.log" 2>&1
rc=$?
set -e
n=$(grep -c "WARNING: ThreadSanitizer" "$OUT/tsan.log" || true)
tail -3 "$OUT/tsan.log"
if [ "$rc" -ne 0 ] || [ "$n" -ne 0 ]; then
  echo "tsan: $n report(s), driver exit code $rc — log in $OUT/tsan.log"
  grep -A12 "WARNING: ThreadSanitizer" "$OUT/tsan.log" | grep -v "std::\|invoke\|libstdc" | head -60
  exit 1
fi
echo "tsan clean ($OUT)"
