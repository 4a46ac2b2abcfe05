#!/bin/bash
# AddressSanitizer + UBSan sweep of every CPU-side component (GPU ASan is not available on the pool):
# the oracle, the product's host stages (msv1_host / sp_host / sp_entropy / sp_models, through the
# test shim) and the stream encoder, driven with valid, truncated, bit-flipped and random streams.
# Usage: tools/sanitize_cpu.sh   (exit 0 and "sanitizer run finished" = clean)
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="${JSP_SANITIZE_DIR:-$(mktemp -d /tmp/jsp_asan.XXXX)}"
export JSP_SANITIZE_DIR="$OUT"
SAN="-O1 -g -std=c++17 -fPIC -shared -fsanitize=address,undefined -fno-sanitize-recover=undefined"
C="$ROOT/jsplayer_amd/csrc"
g++ $SAN -o "$OUT/liboracle.so" "$ROOT"/oracle/*.cpp
g++ $SAN -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Wno-deprecated-declarations -o "$OUT/libhoststage.so" \
    "$ROOT/tests/hoststage/shim.cpp" "$C/msv1_host.cpp" "$C/sp_host.cpp" "$C/sp_entropy.cpp" "$C/sp_models.cpp"
g++ $SAN -DJSP_MODEL_TOOLS -o "$OUT/libjspgen.so" "$ROOT/jsplayer_amd/gen/sp_encoder.cpp" "$C/sp_models.cpp"
LD_PRELOAD="$(g++ -print-file-name=libasan.so) $(g++ -print-file-name=libstdc++.so.6)" \
    ASAN_OPTIONS=detect_leaks=0 python3 "$ROOT/tools/sanitize_cpu.py"
