#!/bin/bash
# lab: a copy of the tree's package under .lab_<name>/ with jsplayer_amd/csrc/<file> replaced by <source> and built with extra flags
#   tools/lab/mk_variant.sh <name> <file in csrc> <source> "<flags>"      (run here: the built .so travels to the GPU box)
set -e
R="$(cd "$(dirname "$0")/../.." && pwd)"; D="$R/.lab_$1"
rm -rf "$D"; mkdir -p "$D"
(cd "$R" && tar -c --exclude='*.o' --exclude='*.so' --exclude=__pycache__ jsplayer_amd include | tar -x -C "$D")
cp "$R/jsplayer_amd/libjspgen.so" "$D/jsplayer_amd/" 2>/dev/null || true
cp "$3" "$D/jsplayer_amd/csrc/$2"
make -C "$D/jsplayer_amd/csrc" -j8 HOOKS="$4" > "$D/make.log" 2>&1 || { tail -20 "$D/make.log"; exit 1; }
rm -f "$D"/jsplayer_amd/csrc/*.o
echo "built $D ($4)"
