#!/bin/bash
# Collect the round's rocprofv3 summaries on a GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1100 -- 'bash scripts/collect_profiles.sh r04'
# Each counter set is its own run; programs are started directly after `--` (no shell hop between rocprofv3 and python).
set -o pipefail
R=${1:-r06}
OUT=gpurun_out/prof_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export HDY_BENCH_SECOND_BLOCK=0   # (round 6: bench.py times a second block of K steps by default; the per-step divisions here count warm-up + K)
export HDY_BENCH_PREWARM_S=0      # the profiled runs count kernels per step: no untimed pre-warm steps in the trace (bench.py)
rocprofv3 --kernel-trace --stats -d $OUT/bench -o bench --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-infer > $OUT/bench.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/roof -o roof --output-format csv -- python3 scripts/roofline_kernel.py > $OUT/roof.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o fetch --output-format csv -- python3 scripts/roofline_kernel.py > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o write --output-format csv -- python3 scripts/roofline_kernel.py > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/mfma -o mfma --output-format csv -- python3 scripts/roofline_kernel.py > $OUT/mfma.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/lds -o lds --output-format csv -- python3 scripts/roofline_kernel.py > $OUT/lds.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/infer -o infer --output-format csv -- python3 scripts/bench_infer.py l 128 1024 3 > $OUT/infer.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/hnet -o hnet --output-format csv -- python3 scripts/bench_hnet.py s 16 1280 4 > $OUT/hnet.log 2>&1
bash scripts/step_profile.sh $R > /dev/null 2>&1 && cp gpurun_out/step_$R/kernel_stats.txt $OUT/step_kernel_stats.txt
python3 scripts/layer_bench.py > $OUT/layer_table.txt 2>&1
python3 scripts/layer_bench.py 32 640 m > $OUT/layer_table_m_b32.txt 2>&1
python3 scripts/layer_bench.py 128 1024 l eval > $OUT/layer_table_l_eval.txt 2>&1
bash scripts/step_profile_variant.sh ${R}m m 32 640 > /dev/null 2>&1 && cp gpurun_out/step_${R}m/kernel_stats.txt $OUT/step_m_kernel_stats.txt
for v in "m 32 640" "l 16 640" "m6 16 1280"; do set -- $v; HDY_BENCH_PREWARM_S=1 python3 bench.py --variant $1 --batch $2 --size $3 --steps 10 --warmup 3 --no-cpu-baseline --no-infer --no-roofline 2>/dev/null | tail -1 | cut -c1-420 >> $OUT/variants.log; done
python3 scripts/bench_mask.py s 16 1280 4 2>/dev/null | tail -1 >> $OUT/variants.log
python3 scripts/bench_latency.py 2>/dev/null | tail -3 >> $OUT/variants.log
PYTHONPATH=. python3 scripts/probes/seg_kernels.py > $OUT/seg_kernels.txt 2>/dev/null
PYTHONPATH=. python3 scripts/probes/dgrad_walk.py > $OUT/dgrad_walk.txt 2>/dev/null
python3 scripts/trace_gaps.py $OUT/bench/bench_kernel_trace.csv 6 > $OUT/trace_gaps.txt 2>&1
python3 scripts/trace_idle.py $OUT/bench/bench_kernel_trace.csv 12 > $OUT/trace_idle.txt 2>&1
python3 scripts/pmc_summary.py $OUT/fetch/fetch_counter_collection.csv $OUT/write/write_counter_collection.csv conv3x3_c64_kernel $OUT/conv3x3_pmc.json > /dev/null 2>&1
python3 scripts/counters_summary.py $OUT/mfma/mfma_counter_collection.csv $OUT/lds/lds_counter_collection.csv conv3x3_c64_kernel $OUT/roof/roof_kernel_stats.csv $OUT/conv3x3_counters.json > /dev/null 2>&1
bash scripts/step_traffic.sh $R > /dev/null 2>&1 && cp gpurun_out/traffic_$R/step_traffic.json $OUT/step_traffic.json
bash scripts/probe_layers_pmc.sh 'B wgrd +128x +256 k3 s2|B wgrd +256x +512 k3 s2|F fwd +128-> +128 k3 s1 @40|F fwd +256-> +256 k1 s1 @40|B bnbw K=256 M=25600|B f1x1 +64<> +64 M=1638400$|B f1x1 +32<> +32|B f1x1 +128<> +128 M=409600|B wgrd +256x +256 k1 s1 @40|B dgrd +128<- +128 k3 s1 @40|B dgrd +512<- +512 k1' ${R}pmc 5 > /dev/null 2>&1 && cp gpurun_out/probe_${R}pmc/table.txt $OUT/layers_pmc_table.txt
ls $OUT $OUT/* | head -60
grep "^{" $OUT/bench.log | tail -1 | cut -c1-300
grep "^{" $OUT/infer.log | tail -1 | cut -c1-300
grep "^{" $OUT/hnet.log | tail -1 | cut -c1-300
cat $OUT/variants.log | cut -c1-300
