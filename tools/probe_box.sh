#!/bin/bash
# Facts about the GPU box that DESIGN.md quotes (toolchains, cores, standalone HIP runtime).
echo "nproc=$(nproc)"; grep -m1 "model name" /proc/cpuinfo
for t in haxe haxelib neko node go javac; do printf "%s: " $t; (command -v $t && $t --version 2>&1 | head -1) || echo absent; done
python - <<'PY'
import ctypes
h = ctypes.CDLL('/opt/rocm/lib/libamdhip64.so')
n = ctypes.c_int()
print("system libamdhip64 alone: hipGetDeviceCount rc", h.hipGetDeviceCount(ctypes.byref(n)), "count", n.value)
PY
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock" | head -8
