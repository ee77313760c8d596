#!/bin/bash
# What the tile boundary of the 256-wide activation instances consists of: C4 shapes on the diagnostic build (scripts/build_variant.sh hdy_deepdbg conv_deep
# -DHDY_DEEP_DBG=1) with HDY_DEEP_DEBUG = 0 (everything), 256 (the epilogue's arithmetic, no stores), 8 (no epilogue); results are wrong under 8 / 256.
cd $GRAFT_REPO_ROOT
for shape in "128 64 64 512 512 1" "128 128 128 256 256 1" "128 64 64 256 256 3"; do
  for rep in 1 2; do for dbg in 0 256 8; do
    echo -n "$shape  HDY_DEEP_DEBUG=$dbg  "; ACT=1 HDY_LIB=libhdy_deepdbg.so HDY_DEEP_DEBUG=$dbg python3 scripts/conv_case_bench.py $shape 2>&1 | grep " us "
  done; done
done
