#!/bin/bash
# round 6 baseline for BASELINE C3's per-GPU step (yolov5m, B = 32): step A/B with the weight-gradient stream on / inline / absent, layer table at B = 32
set -o pipefail
OUT=gpurun_out/r06_base
mkdir -p $OUT
B="python3 bench.py --variant m --batch 32 --steps 20 --warmup 5 --no-cpu-baseline --no-infer --no-roofline"
for i in 1 2; do
  $B 2>/dev/null | tail -1 | cut -c1-200 >> $OUT/m_step.txt
  HDY_SIDE_WGRAD=0 $B 2>/dev/null | tail -1 | cut -c1-200 >> $OUT/m_step_inline.txt
done
HDY_SKIP_WGRAD=1 $B 2>/dev/null | tail -1 | cut -c1-200 >> $OUT/m_step_nowgrad.txt
python3 scripts/layer_bench.py 32 640 m > $OUT/layer_table_m_b32.txt 2>&1
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_s.txt 2>$OUT/bench_s.err
tail -3 $OUT/m_step*.txt; head -20 $OUT/layer_table_m_b32.txt; cut -c1-300 $OUT/bench_s.txt
