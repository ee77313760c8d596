"""Timeline summary of a `rocprofv3 --kernel-trace --output-format csv` of bench.py: per training step, the wall span, the time the GPU
had at least one kernel running, the per-queue busy time and the idle gaps on the busiest queue (the main stream).

    python scripts/trace_gaps.py <kernel_trace.csv> [steps_to_skip]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'], r['Kernel_Name']) for r in rows))
# a step starts at every stem_prep launch
starts = [i for i, e in enumerate(ev) if 'stem_prep' in e[3]]
print(f'{len(ev)} kernels, {len(starts)} steps')
for si in range(skip, len(starts) - 1):
    seg = ev[starts[si]:starts[si + 1]]
    t0, t1 = seg[0][0], seg[-1][1]
    span = (min(t1, ev[starts[si + 1]][0]) - t0) / 1e6
    # union of busy intervals
    busy, cur_s, cur_e = 0, None, None
    for s, e, q, n in seg:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    perq = defaultdict(int)
    for s, e, q, n in seg:
        perq[q] += e - s
    mainq = max(perq, key=perq.get)
    mq = [x for x in seg if x[2] == mainq]
    gaps = [(b[0] - a[1]) for a, b in zip(mq, mq[1:]) if b[0] > a[1]]
    big = sorted(gaps, reverse=True)[:5]
    print(f'step {si}: span {span:.2f} ms, GPU busy (union) {busy/1e6:.2f} ms, per queue ' +
          ', '.join(f'{q}: {v/1e6:.2f} ms / {sum(1 for x in seg if x[2]==q)} k' for q, v in sorted(perq.items(), key=lambda kv: -kv[1])) +
          f'; main-queue gaps: {sum(gaps)/1e6:.2f} ms in {len(gaps)} (largest {[round(g/1e3) for g in big]} us)')
