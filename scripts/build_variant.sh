#!/bin/bash
# build_variant.sh NAME [-DMACRO ...]: csrc/build/libNAME.so = the library with conv3x3.hip recompiled under extra macros (A/B probes)
set -e
R=/root/repo/hd_yolo_amd/csrc
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -I$R -I/root/repo/include "$@" -c $R/conv3x3.hip -o /tmp/c3_$name.o
objs=""
for o in api conv_igemm conv_stem conv_wgrad bn_act pool detect loss roi; do objs="$objs $R/build/$o.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/lib$name.so $objs /tmp/c3_$name.o
echo built $R/build/lib$name.so
