"""Where is the GPU idle inside a training step?  From a `rocprofv3 --kernel-trace --output-format csv` of bench.py: the intervals of one
steady-state step in which NO queue has a kernel running, with the kernels that end before and start after each (scripts/trace_gaps.py gives
the per-step totals).

    python scripts/trace_idle.py <kernel_trace.csv> [top=30]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'], r['Kernel_Name']) for r in rows))
starts = [i for i, e in enumerate(ev) if 'stem_prep' in e[3]]
si = len(starts) * 2 // 3
seg = ev[starts[si]:starts[si + 1] + 1]
t0 = seg[0][0]


def short(n):
    return n.replace('(anonymous namespace)::', '').replace('void ', '').replace('_ZN12_GLOBAL__N_1', '')[:48]


cur_e, last, idle = None, None, []
for s, e, q, n in seg:
    if cur_e is not None and s > cur_e:
        idle.append((cur_e - t0, s - cur_e, last, (n, q)))
    if cur_e is None or e > cur_e:
        cur_e, last = e, (n, q)
mainq = max(set(x[2] for x in seg), key=lambda q: sum(x[1] - x[0] for x in seg if x[2] == q))
after_apply = [d for _, d, l, _ in idle if 'bwd_apply' in l[0]]
print(f'step {si}: span {(seg[-1][0] - t0) / 1e3:.0f} us, no kernel running for {sum(i[1] for i in idle) / 1e3:.0f} us in {len(idle)} intervals '
      f'({sum(after_apply) / 1e3:.0f} us in {len(after_apply)} behind a BatchNorm-backward apply pass = in front of a fork marker)')
for t, d, l, n in sorted(idle, key=lambda x: -x[1])[:top]:
    print(f'  at {t / 1e3:8.1f} us  idle {d / 1e3:6.1f} us  after {short(l[0])} (q{l[1]})  before {short(n[0])} (q{n[1]})')
