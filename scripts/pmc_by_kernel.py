"""Per-kernel averages of a rocprofv3 --pmc counter_collection CSV (all dispatches of a run, e.g. bench.py): kernel name (shortened), grid, dispatches, mean
duration, mean of every counter.  Usage: python scripts/pmc_by_kernel.py <counter_collection.csv> [name regex] [skip first N dispatches per kernel]"""
import csv, re, sys
from collections import OrderedDict, defaultdict
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rows = defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    q = rows[int(r['Dispatch_Id'])]
    q['name'], q['grid'] = r['Kernel_Name'], int(r['Grid_Size'])
    q['us'] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    q.setdefault('c', {})
    q['c'][r['Counter_Name']] = q['c'].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
groups = OrderedDict()
for d in sorted(rows):
    q = rows[d]
    short = re.sub(r'\(anonymous namespace\)::|void |_ZN12_GLOBAL__N_1', '', q['name'])
    short = re.sub(r'\(.*\)$', '', short)[:60]
    if pat and not pat.search(short):
        continue
    groups.setdefault((short, q['grid']), []).append(q)
for (name, grid), qs in groups.items():
    qs = qs[skip:] or qs
    cs = OrderedDict()
    for c in qs[0]['c']:
        cs[c] = sum(q['c'].get(c, 0.0) for q in qs) / len(qs)
    print(f'{name:62s} grid={grid:8d} n={len(qs):4d} us={sum(q["us"] for q in qs) / len(qs):8.1f}  ' + '  '.join(f'{k}={v:.4g}' for k, v in cs.items()))
