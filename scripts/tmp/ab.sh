cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ab -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 > $GRAFT_REPO_ROOT/gpurun_out/ab.log 2>&1
