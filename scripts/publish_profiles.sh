#!/bin/bash
# Copy the summaries that scripts/collect_profiles.sh left under gpurun_out/prof_<round>/ into profiles/<round>_* (the tracked, judged copies):
#   bash scripts/publish_profiles.sh r04
set -e
R=${1:-r06}
O=gpurun_out/prof_$R
grep "^{" $O/bench.log | tail -1 > profiles/${R}_bench.json
cp $O/bench/bench_kernel_stats.csv profiles/${R}_bench_kernel_stats.csv
cp $O/roof/roof_kernel_stats.csv profiles/${R}_roofline_kernel_stats.csv
cp $O/conv3x3_pmc.json profiles/${R}_conv3x3_pmc.json
cp $O/conv3x3_counters.json profiles/${R}_conv3x3_counters.json
python3 - $R <<'PY'
import csv, sys
R = sys.argv[1]
for k in ('fetch', 'write'):
    rows = list(csv.reader(open(f'gpurun_out/prof_{R}/{k}/{k}_counter_collection.csv')))
    ki = rows[0].index('Kernel_Name')
    csv.writer(open(f'profiles/{R}_conv3x3_pmc_{k}_counter_collection.csv', 'w')).writerows([rows[0]] + [r for r in rows[1:] if 'conv3x3_c64_kernel' in r[ki]])
PY
grep "^{" $O/infer.log | tail -1 > profiles/${R}_infer.json; cp $O/infer/infer_kernel_stats.csv profiles/${R}_infer_kernel_stats.csv
grep "^{" $O/hnet.log | tail -1 > profiles/${R}_hnet.json; cp $O/hnet/hnet_kernel_stats.csv profiles/${R}_hnet_kernel_stats.csv
for f in layer_table.txt layer_table_m_b32.txt layer_table_l_eval.txt step_m_kernel_stats.txt step_kernel_stats.txt trace_gaps.txt trace_idle.txt variants.log step_traffic.json layers_pmc_table.txt seg_kernels.txt; do cp $O/$f profiles/${R}_$f; done
{ echo "# PYTHONPATH=. python scripts/probes/dgrad_walk.py (B = 64, the yolov5s bench shapes; deep pipeline with HDY_DEEP_WALK=1 | generic kernel's walk)"; cat $O/dgrad_walk.txt; } > profiles/${R}_dgrad_walk.txt
ls -la profiles/${R}_* | wc -l
