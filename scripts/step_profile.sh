#!/bin/bash
# In-step per-kernel totals of the C2 train step (no roofline / inference / CPU legs in the process):
#   gpurun --timeout 600 -- 'bash scripts/step_profile.sh tag'   ->  gpurun_out/step_<tag>/kernel_stats.txt
set -o pipefail
TAG=${1:-a}
OUT=$GRAFT_REPO_ROOT/gpurun_out/step_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export HDY_BENCH_SECOND_BLOCK=0   # (round 6: bench.py times a second block of K steps by default; the per-step divisions here count warm-up + K)
export HDY_BENCH_PREWARM_S=0      # the profiled runs count kernels per step: no untimed pre-warm steps in the trace (bench.py)
rocprofv3 --kernel-trace --stats -d $OUT/p -o p --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-infer --no-roofline --no-cpu-baseline > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | cut -c1-160
python3 - $OUT <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + '/p/**/p_kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
steps = 13
out = open(sys.argv[1] + '/kernel_stats.txt', 'w')
tot = 0
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs'])):
    ms = float(r['TotalDurationNs']) / steps / 1e6
    tot += ms
    line = f"{ms:7.3f} ms/step  calls/step {int(r['Calls'])/steps:6.1f}  avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:100]}"
    print(line, file=out)
print(f'total kernel time per step {tot:.2f} ms', file=out)
out.close()
print(open(sys.argv[1] + '/kernel_stats.txt').read()[:3500])
PY
