#!/bin/bash
cd $GRAFT_REPO_ROOT
for L in "$@"; do for mode in "128" "128 stats" "256"; do HDY_LIB=$L python3 scripts/probes/tile_boundary_fit.py $mode 2>&1 | grep -v amdgpu.ids; done; done
