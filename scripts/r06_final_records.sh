#!/bin/bash
# Round 6, final HEAD: the bench line under the profiler (kernel stats + trace -> gaps / step order) and the driver's command unprofiled, twice.
set -o pipefail
OUT=gpurun_out/final_r06
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
HDY_BENCH_SECOND_BLOCK=0 HDY_BENCH_PREWARM_S=0 rocprofv3 --kernel-trace --stats -d $OUT/bench -o bench --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-infer > $OUT/bench.log 2>&1
python3 scripts/trace_gaps.py $OUT/bench/bench_kernel_trace.csv 6 > $OUT/trace_gaps.txt 2>&1
python3 scripts/trace_idle.py $OUT/bench/bench_kernel_trace.csv 12 > $OUT/trace_idle.txt 2>&1
python3 scripts/trace_step_order.py $OUT/bench/bench_kernel_trace.csv > $OUT/step_order.txt 2>&1
python3 bench.py --steps 20 --warmup 5 > $OUT/driver1.json 2>$OUT/driver1.err
python3 bench.py --steps 20 --warmup 5 > $OUT/driver2.json 2>$OUT/driver2.err
python3 bench.py --variant m --batch 32 --steps 20 --warmup 5 --no-cpu-baseline --no-infer --no-roofline > $OUT/m1.json 2>/dev/null
cut -c1-260 $OUT/driver1.json $OUT/driver2.json $OUT/m1.json; tail -3 $OUT/trace_gaps.txt
