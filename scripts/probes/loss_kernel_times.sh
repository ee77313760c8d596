cd $GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for L in libhdyolo_hip.so libhdy_oldloss.so; do
  rm -rf gpurun_out/lt; HDY_LIB=$L HDY_BENCH_PREWARM_S=0 rocprofv3 --kernel-trace --stats -d gpurun_out/lt -o p --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-infer --no-roofline --no-cpu-baseline > /dev/null 2>&1
  f=$(ls gpurun_out/lt/p_kernel_stats.csv gpurun_out/lt/*/p_kernel_stats.csv 2>/dev/null | head -1)
  echo "== $L"; grep -E "dense_kernel|match_kernel" $f | awk -F, '{print $1, $2, $4}' | cut -c1-120
done
