"""Per-layer table of rocprofv3 --pmc counter_collection CSVs collected on `HDY_PROBE_MARKERS=1 python scripts/layer_probe.py <regex> <reps>`:
the int16 fill kernels the probe launches before every matched record cut the dispatch sequence into segments (segment i = the i-th label
the probe printed); inside a segment every kernel's LAST reps/(reps + 2) dispatches are averaged per counter (the first two replays are warm-up).
Usage: python scripts/pmc_layers.py <reps> <time.log of the probe> a_counter_collection.csv [b_counter_collection.csv ...]"""
import csv, re, sys
from collections import OrderedDict, defaultdict

reps = int(sys.argv[1])
labels = [l[:44].strip() for l in open(sys.argv[2]) if re.match(r'[FB] ', l)]
table = OrderedDict()                                    # (segment, kernel) -> counter -> value
for path in sys.argv[3:]:
    rows = defaultdict(dict)                             # dispatch -> {name, grid, dur, counters}
    for r in csv.DictReader(open(path)):
        d = int(r['Dispatch_Id'])
        q = rows[d]
        q['name'], q['grid'], q['wg'] = r['Kernel_Name'], int(r['Grid_Size']), int(r['Workgroup_Size'])
        q['vgpr'], q['lds'] = int(r['VGPR_Count']) + int(r['Accum_VGPR_Count']), int(r.get('LDS_Block_Size', 0) or 0)
        q['us'] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        q.setdefault('c', {})
        q['c'][r['Counter_Name']] = q['c'].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    seg, per = -1, defaultdict(list)
    for d in sorted(rows):
        q = rows[d]
        if 'FillFunctor<short>' in q['name'] or 'FillFunctor<int16' in q['name']:
            seg += 1                                     # one marker per matched record, in the order of the probe's output
            continue
        if seg >= 0:
            short = re.sub(r'\(anonymous namespace\)::|void |_ZN12_GLOBAL__N_1', '', q['name'])
            short = re.sub(r'\(.*\)$', '', short)[:70]
            per[(seg, short, q['grid'], q['wg'], q['vgpr'], q['lds'])].append(q)
    remap = {s: s for s in {k[0] for k in per}}
    for key, qs in per.items():
        n = len(qs) * reps // (reps + 2)
        qs = qs[-n:] if n else qs
        row = table.setdefault((remap[key[0]],) + key[1:], OrderedDict())
        row.setdefault('us', sum(q['us'] for q in qs) / len(qs))
        row['n'] = len(qs)
        for c in qs[0]['c']:
            row[c] = sum(q['c'].get(c, 0.0) for q in qs) / len(qs)
for key, row in sorted(table.items(), key=lambda kv: kv[0][0]):
    lab = labels[key[0]] if key[0] < len(labels) else f'segment {key[0]}'
    print(f'[{lab}]  {key[1]}  grid={key[2]} wg={key[3]} regs={key[4]} lds={key[5]}')
    print('    ' + '  '.join(f'{k}={v:.4g}' for k, v in row.items()))
