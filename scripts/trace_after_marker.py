"""Average duration of the kernels replayed by `HDY_PROBE_MARKERS=1 scripts/layer_probe.py` from a rocprofv3 kernel-trace CSV: per marker
segment (one per matched record) the LAST `reps` dispatches of every kernel whose name matches the regex.
Usage: python scripts/trace_after_marker.py <kernel_trace.csv> <reps> <name-regex> [tag]"""
import csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
reps, pat, tag = int(sys.argv[2]), re.compile(sys.argv[3]), (sys.argv[4] if len(sys.argv) > 4 else '')
seg, per = -1, {}
for r in rows:
    n = r['Kernel_Name']
    if 'FillFunctor<short>' in n or 'FillFunctor<int16' in n:
        seg += 1
        continue
    if seg >= 0 and pat.search(n):
        short = re.sub(r'\(anonymous namespace\)::|void |_ZN12_GLOBAL__N_1|\(.*\)$', '', n)[:48]
        per.setdefault((seg, short), []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for (seg, short), v in sorted(per.items()):
    v = v[-reps:]
    print(f'{tag:>8s} seg {seg}  avg {sum(v)/len(v):7.1f} us  min {min(v):7.1f}  n={len(v)}  {short}')
