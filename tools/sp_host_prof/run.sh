#!/bin/bash
# Times (and with PG=1 gprof-profiles) the ScreenPressor HOST stage alone on the CPU it runs on: no GPU involved.
# usage: tools/sp_host_prof/run.sh [outdir]      (from the repo root)
set -e
OUT=${1:-gpurun_out/sp_host_prof}
mkdir -p $OUT
S=jsplayer_amd/csrc
CXX=/opt/rocm/lib/llvm/bin/clang++   # what the library's host code is built with (hipcc)
FLAGS="-O3 -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I$S -Wno-deprecated-declarations -Wno-unused-result"
$CXX $FLAGS -o $OUT/prof tools/sp_host_prof/prof.cpp $S/sp_host.cpp $S/sp_entropy.cpp $S/sp_models.cpp
$CXX $FLAGS -pg -o $OUT/prof_pg tools/sp_host_prof/prof.cpp $S/sp_host.cpp $S/sp_entropy.cpp $S/sp_models.cpp
B=tools/sp_host_prof/base   # optional: an older copy of the sources to time next to the current ones
[ -d $B ] && $CXX -O3 -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I$B -Wno-deprecated-declarations -Wno-unused-result -o $OUT/prof_base tools/sp_host_prof/prof.cpp $B/sp_host.cpp $B/sp_entropy.cpp $B/sp_models.cpp
for w in screenpressor_v4_1080p_iframes:8 screenpressor_v2_1080p_iframes:4 screenpressor_v4_1080p_pclip300:60; do
    name=${w%%:*}; n=${w##*:}
    [ -f $OUT/$name.bin ] || PYTHONPATH=. python tools/sp_host_prof/dump_frames.py $name $n $OUT/$name.bin
    echo "== $name"
    [ -x $OUT/prof_base ] && { echo -n "base: "; taskset -c 3 $OUT/prof_base $OUT/$name.bin 6; }
    echo -n "now:  "; taskset -c 3 $OUT/prof $OUT/$name.bin 6
    (cd $OUT && ./prof_pg $name.bin 4 > /dev/null && gprof -b -p prof_pg gmon.out | head -16)
done
