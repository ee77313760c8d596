# timing ablations of the deep-pipelined conv kernel by kernel-trace durations (HDY_DEEP_DEBUG bits: 1 no A loads, 2 no B loads, 4 no MFMAs,
# 8 no epilogue, 16 no fragment reads)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; export HDY_LIB=libhdy_deepdbg.so
OUT=$GRAFT_REPO_ROOT/gpurun_out/abl
rm -rf $OUT; mkdir -p $OUT
export HDY_PROBE_MARKERS=1
PAT=${1:-'F fwd +128-> +128 k3 s1 @40|F fwd +256-> +256 k1 s1 @40|B dgrd +128<- +128 k3 s1 @40|B dgrd +512<- +512 k1|B dgrd +256<- +256 k1 s1 @40'}
for d in ${2:-0 1 2 3 4 8 16 20 23 31 g}; do
  if [ $d = g ]; then export HDY_NO_DEEP=1; export HDY_DEEP_DEBUG=0; else unset HDY_NO_DEEP; export HDY_DEEP_DEBUG=$d; fi
  rocprofv3 --kernel-trace -d $OUT/r -o r --output-format csv -- python3 scripts/layer_probe.py "$PAT" 10 > $OUT/log_$d.txt 2>&1
  f=$(ls $OUT/r/*/r_kernel_trace.csv $OUT/r/r_kernel_trace.csv 2>/dev/null | head -1)
  python3 scripts/trace_after_marker.py "$f" 10 'conv_deep|conv_igemm' "dbg=$d" >> $OUT/abl.txt
  rm -rf $OUT/r
done
grep "^[FB] " $OUT/log_0.txt | cut -c1-44 | nl -v0
sort -k3,3n -s $OUT/abl.txt
