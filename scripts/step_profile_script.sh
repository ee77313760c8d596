#!/bin/bash
# Per-kernel totals of any of the step scripts under rocprofv3 (run through gpurun from the repo root):
#   gpurun --timeout 600 -- 'bash scripts/step_profile_script.sh tag STEPS_IN_TRACE scripts/bench_mask.py s 16 1280 10'
# STEPS_IN_TRACE = warm-up + timed steps the script runs (its per-step figures divide by it).
set -o pipefail
TAG=$1; STEPS=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/step_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $OUT/p -o p --output-format csv -- python3 "$@" > $OUT/run.log 2>&1
tail -1 $OUT/run.log | cut -c1-200
python3 - $OUT $STEPS <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + '/p/**/p_kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
steps = int(sys.argv[2])
out = open(sys.argv[1] + '/kernel_stats.txt', 'w')
tot = 0
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs'])):
    ms = float(r['TotalDurationNs']) / steps / 1e6
    tot += ms
    print(f"{ms:7.3f} ms/step  calls/step {int(r['Calls'])/steps:6.1f}  avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:110]}", file=out)
print(f'total kernel time per step {tot:.2f} ms', file=out)
out.close()
print(open(sys.argv[1] + '/kernel_stats.txt').read()[:6000])
PY
