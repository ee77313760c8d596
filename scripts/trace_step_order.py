"""One steady-state training step of a `rocprofv3 --kernel-trace --output-format csv` of bench.py, kernel by kernel in start order:
start offset from the step's first kernel, duration, hardware queue, idle time on that queue in front of the kernel, short name.

    python scripts/trace_step_order.py <kernel_trace.csv> [step index, default: the last but one]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'], r['Kernel_Name']) for r in rows))
starts = [i for i, e in enumerate(ev) if 'stem_prep' in e[3]]
si = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) - 2
seg = ev[starts[si]:starts[si + 1]]
t0 = seg[0][0]
last_end = {}
qn = {}


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'_ZN12_GLOBAL__N_1\d+', '', n)
    return n[:70]


print(f'step {si}: {len(seg)} kernels, span {(seg[-1][1] - t0) / 1e3:.1f} us')
for s, e, q, n in seg:
    qi = qn.setdefault(q, len(qn))
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = max(e, last_end.get(q, 0))
    print(f'{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:7.1f} us  q{qi}  gap {gap:6.1f}  {short(n)}')
