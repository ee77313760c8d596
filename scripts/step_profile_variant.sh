#!/bin/bash
# In-step per-kernel totals of a bench.py variant:   bash scripts/step_profile_variant.sh <tag> <variant> <batch> <size>   ->  gpurun_out/step_<tag>/kernel_stats.txt
set -o pipefail
TAG=${1:-m}; V=${2:-m}; B=${3:-32}; S=${4:-640}
OUT=$GRAFT_REPO_ROOT/gpurun_out/step_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
HDY_BENCH_SECOND_BLOCK=0 HDY_BENCH_PREWARM_S=0 rocprofv3 --kernel-trace --stats -d $OUT/p -o p --output-format csv -- python3 bench.py --variant $V --batch $B --size $S --steps 10 --warmup 3 --no-infer --no-roofline --no-cpu-baseline > $OUT/bench.log 2>&1
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
    print(f"{ms:7.3f} ms/step  calls/step {int(r['Calls'])/steps:6.1f}  avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:110]}", file=out)
print(f'total kernel time per step {tot:.2f} ms', file=out)
out.close()
PY
rm -rf $OUT/p
