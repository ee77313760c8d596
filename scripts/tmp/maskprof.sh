cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/mk -- python3 $GRAFT_REPO_ROOT/scripts/bench_mask.py s 16 1280 6 > $GRAFT_REPO_ROOT/gpurun_out/mk.log 2>&1
