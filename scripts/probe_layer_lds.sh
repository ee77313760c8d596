#!/bin/bash
# LDS / MFMA occupancy counters of single layers of the C2 train plan (run through gpurun from the repo root):
#   gpurun --timeout 900 -- 'bash scripts/probe_layer_lds.sh "F fwd +128-> +128 k3 s1" tag'
set -o pipefail
PAT="$1"; TAG=${2:-lds}; REPS=${3:-5}
OUT=$GRAFT_REPO_ROOT/gpurun_out/probe_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o p$i --output-format csv -- python3 scripts/layer_probe.py "$PAT" $REPS > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; }
done
python3 scripts/pmc_table.py $((REPS)) "${4:-conv_igemm}" $(ls $OUT/p*/p*_counter_collection.csv $OUT/p*/*/p*_counter_collection.csv 2>/dev/null) > $OUT/table.txt; cut -c1-1200 $OUT/table.txt
