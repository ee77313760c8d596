#!/bin/bash
# Whole-step HBM traffic of the C2 train step: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE runs (the guide's rule) over
# `bench.py --steps 10 --warmup 3` (program directly after `--`), summed over every dispatch and divided by the 13 steps.
#   gpurun --timeout 600 -- 'bash scripts/step_traffic.sh r03'  ->  gpurun_out/traffic_<tag>/step_traffic.json (copy to profiles/<tag>_step_traffic.json)
set -o pipefail
TAG=${1:-r03}
OUT=$GRAFT_REPO_ROOT/gpurun_out/traffic_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export HDY_BENCH_SECOND_BLOCK=0   # (round 6: bench.py times a second block of K steps by default; the per-step divisions here count warm-up + K)
export HDY_BENCH_PREWARM_S=0      # the profiled runs count kernels per step: no untimed pre-warm steps in the trace (bench.py)
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/f -o f --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-infer --no-roofline --no-cpu-baseline > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/w -o w --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-infer --no-roofline --no-cpu-baseline > $OUT/write.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, json, re, sys
out = sys.argv[1]
def total(kind, name):
    f = glob.glob(f'{out}/{kind}/**/{kind}_counter_collection.csv', recursive=True)[0]
    tot, per, calls = 0.0, {}, {}
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == name:
            v = float(r['Counter_Value'])
            tot += v
            k = re.sub(r'\(.*$', '', re.sub(r'void |\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d+', '', r['Kernel_Name']))[:48]
            per[k] = per.get(k, 0.0) + v
            calls[k] = calls.get(k, 0) + 1
    CALLS.update(calls)
    return tot, per
steps = 13
CALLS = {}
fetch_kb, pf = total('f', 'FETCH_SIZE')
write_kb, pw = total('w', 'WRITE_SIZE')
line = [l for l in open(f'{out}/fetch.log') if l.startswith('{')][-1]
algo = json.loads(line)['step']['algorithmic_gb_per_step']
# gfx950: FETCH_SIZE counts 64 B per 128-B request of a wide streaming read -> doubled (MI355X_MICROARCH.md, HBM); WRITE_SIZE is exact
res = {'steps': steps, 'fetch_gb_per_step_raw': round(fetch_kb * 1024 / steps / 1e9, 2), 'fetch_gb_per_step': round(2 * fetch_kb * 1024 / steps / 1e9, 2),
       'write_gb_per_step': round(write_kb * 1024 / steps / 1e9, 2), 'algorithmic_gb_per_step': algo}
res['hbm_gb_per_step'] = round(res['fetch_gb_per_step'] + res['write_gb_per_step'], 2)
res['traffic_ratio'] = round(res['hbm_gb_per_step'] / algo, 3)
res['note'] = 'FETCH_SIZE doubled (gfx950 half-count of wide streaming reads); narrow gathers may be over-corrected: ratio is an upper bound'
top = sorted(((2 * pf.get(k, 0) + pw.get(k, 0)) * 1024 / steps / 1e9, k) for k in set(pf) | set(pw))[::-1][:12]
res['top_kernels_gb_per_step'] = [[round(v, 2), k] for v, k in top]
# round 6: the whole per-kernel table — fetch (doubled) and write separately, launches per step
res['kernels'] = [{'kernel': k, 'fetch_gb_per_step': round(2 * pf.get(k, 0) * 1024 / steps / 1e9, 3), 'write_gb_per_step': round(pw.get(k, 0) * 1024 / steps / 1e9, 3),
                   'launches_per_step': round(CALLS.get(k, 0) / steps, 1)} for _, k in sorted(((2 * pf.get(k, 0) + pw.get(k, 0)), k) for k in set(pf) | set(pw))[::-1]]
json.dump(res, open(f'{out}/step_traffic.json', 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
