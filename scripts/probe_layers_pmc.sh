#!/bin/bash
# PMC passes on SEVERAL layers of the C2 train plan in one go (run through gpurun from the repo root):
#   gpurun --timeout 900 -- 'bash scripts/probe_layers_pmc.sh "B wgrd +128x +256 k3 s2|F fwd +128-> +128 k3 s1 @40" tag'
# One counter set per run; python3 directly after `--`; the probe's marker kernels cut the counter CSVs into per-layer segments.
set -o pipefail
PAT="$1"; TAG=${2:-layers}; REPS=${3:-5}
OUT=$GRAFT_REPO_ROOT/gpurun_out/probe_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export HDY_PROBE_MARKERS=1
python3 scripts/layer_probe.py "$PAT" $REPS > $OUT/time.log 2>&1 || { tail -5 $OUT/time.log; exit 1; }
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_VALU" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o p$i --output-format csv -- python3 scripts/layer_probe.py "$PAT" $REPS > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; }
done
grep -v amdgpu.ids $OUT/time.log
python3 scripts/pmc_layers.py $REPS $OUT/time.log $(ls $OUT/p*/p*_counter_collection.csv $OUT/p*/*/p*_counter_collection.csv 2>/dev/null) > $OUT/table.txt; cut -c1-1500 $OUT/table.txt
