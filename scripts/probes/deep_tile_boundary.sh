#!/bin/bash
# Where do conv_deep's short-K layers lose their time at tile boundaries?  (round 4)  Debug build (scripts/build_variant.sh hdy_deepdbg conv_deep -DHDY_DEEP_DBG=1):
#   HDY_DEEP_DEBUG 0 baseline, 8 no epilogue, 256 epilogue arithmetic without its stores, 160 = stamps + cycles per K-tile by position after an epilogue
cd $GRAFT_REPO_ROOT
export HDY_LIB=libhdy_deepdbg.so
PAT='F fwd +256-> +256 k1 s1 @40|F fwd +128-> +128 k3 s1 @40|B dgrd +256<- +256 k1 s1 @40|F fwd +512-> +512 k1|F fwd +1024|B dgrd +512<- +512 k1|F fwd +128-> +128 k1 s1 @80'
for d in 0 8 256 160; do
  echo "== HDY_DEEP_DEBUG=$d"
  HDY_DEEP_DEBUG=$d python3 scripts/deep_stamps.py "$PAT" 2>&1 | grep -v amdgpu.ids
done
