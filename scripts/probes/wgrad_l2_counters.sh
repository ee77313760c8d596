#!/bin/bash
# VERDICT r05 item 4, first step: are the generic weight gradient's operands re-fetched across XCDs?  L2 (TCC) hits / misses / fabric read requests of
# wgrad_kernel (a) replayed alone on three layers of the C2 plan, (b) over whole train steps (under --pmc the dispatches run one after the other: the counters
# show what each launch finds in and asks of L2, not the concurrent case).  Program directly after `--`.
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/wgrad_l2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export HDY_PROBE_MARKERS=1
PAT='B wgrd +256x +256 k1 s1 @40|B wgrd +512x +512 k1 s1 @20|B wgrd +256x +128 k1 s1 @80|B wgrd +128x +128 k3 s1 @40'
python3 scripts/layer_probe.py "$PAT" 5 > $OUT/time.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum -d $OUT/alone -o alone --output-format csv -- python3 scripts/layer_probe.py "$PAT" 5 > $OUT/alone.log 2>&1
python3 scripts/pmc_layers.py 5 $OUT/time.log $(ls $OUT/alone/alone_counter_collection.csv $OUT/alone/*/alone_counter_collection.csv 2>/dev/null) > $OUT/alone_table.txt
unset HDY_PROBE_MARKERS
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum -d $OUT/step -o step --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-infer --no-roofline --no-cpu-baseline > $OUT/step.log 2>&1
python3 scripts/pmc_by_kernel.py $(ls $OUT/step/step_counter_collection.csv $OUT/step/*/step_counter_collection.csv 2>/dev/null | head -1) 'wgrad' 2 > $OUT/step_table.txt
grep -v amdgpu.ids $OUT/time.log; cat $OUT/alone_table.txt; cat $OUT/step_table.txt
