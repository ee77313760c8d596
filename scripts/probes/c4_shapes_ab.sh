#!/bin/bash
# C4 layer shapes (yolov5l, batch 128, 1024x1024) under several builds of the library, SiLU epilogue and raw:  bash scripts/probes/c4_shapes_ab.sh lib1.so lib2.so ...
cd $GRAFT_REPO_ROOT
for shape in "128 64 64 512 512 1" "128 128 128 256 256 1" "128 32 32 1024 1024 1" "128 64 64 256 256 3" "128 128 128 128 128 3"; do
  for L in "$@"; do for act in 1 0; do
    echo -n "$shape  $L act=$act  "; ACT=$act HDY_LIB=$L python3 scripts/conv_case_bench.py $shape 2>&1 | grep " us "
  done; done
done
