#!/bin/bash
# lab: the ScreenPressor key-frame kernel taken apart on the frames of the product's placed pool — whole, everything but the row stores
# (-DJSP_SP_LAB_NOSTORE), nothing but the row stores (-DJSP_SP_LAB_STOREONLY), the row stores without the record-window fetches between
# them (-DJSP_SP_LAB_NOFETCH) — same call, alternating.  step ms | frac of 8 TB/s | pool probe
R="${GRAFT_REPO_ROOT:-$(pwd)}"
build() {  # $1 = dir, $2 = extra flags
  rm -rf $1 && mkdir $1 && cp -r $R/jsplayer_amd $R/include $R/bench.py $R/tests $R/oracle $R/profiles $R/__graft_entry__.py $1/ 2>/dev/null
  (cd $1/jsplayer_amd/csrc && rm -f sp_kernels.o && make CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-function --offload-arch=gfx950 $2" > $1/make.log 2>&1 || tail -5 $1/make.log)
}
build /tmp/alt_nostore "-DJSP_SP_LAB_NOSTORE"
build /tmp/alt_storeonly "-DJSP_SP_LAB_STOREONLY"
build /tmp/alt_nofetch "-DJSP_SP_LAB_STOREONLY -DJSP_SP_LAB_NOFETCH"
build /tmp/alt_cached "-DJSP_SP_LAB_CACHED_RECORDS"
one() { (cd $1 && python bench.py --workload screenpressor_v4_1080p_iframes --steps 30 --warmup 5 --no-e2e --no-cpu-baseline --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['config']['destination_frames']['probe_GBs'])"); }
for i in 1 2; do
  echo -n "whole      "; one $R
  echo -n "no stores  "; one /tmp/alt_nostore
  echo -n "stores only"; one /tmp/alt_storeonly
  echo -n "stores only, no record fetches in the loop"; one /tmp/alt_nofetch
  echo -n "whole, every record window read out of the same 8 KB (cache hits; pixels garbage)"; one /tmp/alt_cached
done
