"""Shader-clock stamps of the deep-pipelined conv kernel on single layers of the C2 train plan (HDY_DEEP_DEBUG bit 32; other bits = the
timing ablations of conv_deep.hip).  Usage: HDY_DEEP_DEBUG=32 python scripts/deep_stamps.py '<layer regex>'"""
import ctypes, os, re, sys
os.environ.setdefault('YOLOv5_VERBOSE', 'false')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hd_yolo_amd import _lib, ops, synth
from hd_yolo_amd.bench_util import describe, flat_records, time_record
from metayolo.models.yolo import Model

pat = re.compile(sys.argv[1])
m = Model(synth.make_cfg('s', 8), synth.make_hyp())
m.load_state_dict(synth.synth_state_dict(synth.shapes_of(m), seed=0), strict=False)
m = m.to('cuda:0').train(); m.half()
x = synth.synth_images(64, 640, seed=0).to('cuda:0')
t = synth.synth_targets(64, 640, 8, seed=1)
l, _ = m(x, t); l['det']['det_loss'].backward()
plan = next(iter(m._eng().plans.values()))
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_ulonglong * (256 * 8))()
sbuf = (ctypes.c_ulonglong * (256 * 16))()
seen = set()
for ph, recs in (('F', plan.fwd), ('B', plan.bwd)):
    for rec in flat_records(recs):
        d, fl, by = describe(rec)
        label = f'{ph} {d}'
        if not pat.search(label) or label in seen:
            continue
        seen.add(label)
        us = time_record(rec, 10)
        torch.cuda.synchronize()
        assert lib.hdy_deep_debug_read(buf) == 0
        rows = [[buf[w * 8 + i] for i in range(8)] for w in range(256)]
        rows = [r for r in rows if r[2] > 0]
        big = max(r[2] for r in rows)
        full = [r for r in rows if r[2] == big]
        avg = lambda i: sum(r[i] for r in full) / len(full)
        print(f'{label:40s} {us:7.1f} us  dbg={os.environ.get("HDY_DEEP_DEBUG")}  wgs {len(rows)} (longest walk: {len(full)} with {big} phases)  '
              f'wave0: loop {avg(0) / big:7.0f} cyc/phase, epilogue {avg(1):8.0f} cyc total, kernel {avg(3):8.0f} cyc   '
              f'wave4: loop {avg(4) / big:7.0f} cyc/phase, kernel {avg(7):8.0f} cyc  -> clock {avg(3) / us / 1e3:5.2f} GHz if the kernel were the whole time', flush=True)
        if int(os.environ.get('HDY_DEEP_DEBUG', '0')) & 128:
            kbuf = (ctypes.c_ulonglong * (256 * 8))()
            assert lib.hdy_deep_debug_read_ktiles(kbuf) == 0
            tot = [sum(kbuf[w * 8 + i] for w in range(256)) for i in range(8)]
            print('      K-tile position after an epilogue -> cycles per K-tile: ' + '  '.join(
                f'{("kt0", "kt1", "kt2", "later")[i]} {tot[i] / max(tot[4 + i], 1):7.0f} (n={tot[4 + i]})' for i in range(4)), flush=True)
        if int(os.environ.get('HDY_DEEP_DEBUG', '0')) & 64:
            assert lib.hdy_deep_debug_read_segments(sbuf) == 0
            idx = [w for w in range(256) if buf[w * 8 + 2] == big]
            names = ['issue', 'vmcnt wait', 'lgkmcnt wait', 'barrier 1', 'MFMAs', 'reads + barrier 2']
            for g in (0, 1):
                tot = [sum(sbuf[w * 16 + g * 8 + i] for w in idx) / len(idx) / big for i in range(6)]
                print(f'      wave {4 * g}: cycles per phase  ' + '  '.join(f'{n} {v:6.0f}' for n, v in zip(names, tot)) + f'   sum {sum(tot):6.0f}', flush=True)
