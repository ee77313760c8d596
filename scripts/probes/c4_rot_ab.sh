#!/bin/bash
# C4 layer shapes with the rotated epilogue (default) and without (HDY_DEEP_DEBUG=512), SiLU epilogue, alternating:  bash scripts/probes/c4_rot_ab.sh
cd $GRAFT_REPO_ROOT
for shape in "128 64 64 512 512 1" "128 128 128 256 256 1" "128 32 32 1024 1024 1" "128 64 64 256 256 3" "128 32 32 512 512 3"; do
  for rep in 1 2; do for dbg in 0 512; do
    echo -n "$shape  HDY_DEEP_DEBUG=$dbg  "; ACT=1 HDY_DEEP_DEBUG=$dbg python3 scripts/conv_case_bench.py $shape 2>&1 | grep " us "
  done; done
done
