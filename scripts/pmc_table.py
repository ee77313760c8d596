"""Per-kernel table of rocprofv3 --pmc counter_collection CSVs: for every (kernel, grid, workgroup) the LAST `keep` dispatches are
averaged per counter (the earlier ones are the plan's own step and the warm-up replays).
Usage: python scripts/pmc_table.py <keep> <name-regex> a_counter_collection.csv [b_counter_collection.csv ...]"""
import csv, re, sys
from collections import OrderedDict, defaultdict

keep = int(sys.argv[1])
pat = re.compile(sys.argv[2])
table = OrderedDict()
for path in sys.argv[3:]:
    per = defaultdict(lambda: defaultdict(dict))           # key -> counter -> dispatch -> value
    dur = defaultdict(dict)
    for r in csv.DictReader(open(path)):
        name = r['Kernel_Name']
        if not pat.search(name):
            continue
        short = re.sub(r'void \(anonymous namespace\)::', '', name)
        short = re.sub(r'\(.*\)$', '', short)
        key = (short, int(r['Grid_Size']), int(r['Workgroup_Size']), int(r['VGPR_Count']), int(r['Accum_VGPR_Count']))
        d = int(r['Dispatch_Id'])
        per[key][r['Counter_Name']][d] = per[key][r['Counter_Name']].get(d, 0.0) + float(r['Counter_Value'])
        dur[key][d] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    for key, counters in per.items():
        row = table.setdefault(key, OrderedDict())
        ds = sorted(dur[key])[-keep:]
        row.setdefault('us', sum(dur[key][d] for d in ds) / len(ds))
        row['n'] = len(ds)
        for c, vals in counters.items():
            row[c] = sum(vals[d] for d in ds) / len(ds)
for key, row in table.items():
    print(f'{key[0]}  grid={key[1]} wg={key[2]} vgpr={key[3]}+{key[4]}')
    print('    ' + '  '.join(f'{k}={v:.4g}' for k, v in row.items()))
