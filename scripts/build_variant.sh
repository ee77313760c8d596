#!/bin/bash
# build_variant.sh NAME SRC [-DMACRO ...]: csrc/build/libNAME.so = the library with csrc/SRC.hip recompiled under extra macros (A/B probes,
# diagnostic builds); load it with HDY_LIB=libNAME.so.  Example: scripts/build_variant.sh hdy_deepdbg conv_deep -DHDY_DEEP_DBG=1
set -e
R=$(cd "$(dirname "$0")/.." && pwd)/hd_yolo_amd/csrc
name=$1; src=$2; shift 2
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -I$R -I$R/../../include "$@" -c $R/$src.hip -o /tmp/var_$name.o
objs=""
for o in $R/build/*.o; do [ "$(basename $o)" = "$src.o" ] || objs="$objs $o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/lib$name.so $objs /tmp/var_$name.o
echo built $R/build/lib$name.so
