"""Per-launch timing of the training plan (yolov5s, B=64, 640, bf16): every record of the forward and backward
launch lists is replayed `reps` times between HIP events; prints time, algorithmic FLOP/s and bytes/s per launch."""
import os, sys, json
os.environ.setdefault('YOLOv5_VERBOSE', 'false')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hd_yolo_amd import synth, ops
from metayolo.models.yolo import Model

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
S = int(sys.argv[2]) if len(sys.argv) > 2 else 640
variant = sys.argv[3] if len(sys.argv) > 3 else 's'
reps = 10
EVAL = len(sys.argv) > 4 and sys.argv[4] == 'eval'     # time the eval launch list (BN folded, SiLU in the conv epilogue)
m = Model(synth.make_cfg(variant, 8), synth.make_hyp())
m.load_state_dict(synth.synth_state_dict(synth.shapes_of(m), seed=0), strict=False)
m = m.to('cuda:0').train(); m.half()
x = synth.synth_images(B, S, seed=0).to('cuda:0')
if EVAL:
    m.eval()
    with torch.no_grad():
        plan = m._eng().plan_for(x, False, torch.bfloat16)
        plan.run_forward(x)
else:
    t = synth.synth_targets(B, S, 8, seed=1)
    l, _ = m(x, t); l['det']['det_loss'].backward()
    plan = next(iter(m._eng().plans.values()))
torch.cuda.synchronize()

from hd_yolo_amd.bench_util import describe, time_record  # noqa: E402

rows = []
for phase, recs in (('F', plan.fwd), ('B', plan.bwd or [])):
    flat = []
    for rec in recs:                      # side-stream brackets: time the bracketed launches themselves, on the main stream
        if rec[0] == '@fork':
            flat.extend(rec[2])
        elif rec[0][0] != '@':
            flat.append(rec)
    for rec in flat:
        us = time_record(rec, reps)          # the same measurement as bench.py's `layers_3x3` rows: fastest of 5 blocks of `reps`
        d, fl, by = describe(rec)
        rows.append((phase, d, us, fl, by))
tot = sum(r[2] for r in rows)
print(f'sum of launches: {tot/1e3:.2f} ms')
agg = {}
for ph, d, us, fl, by in rows:
    k = d.split()[0]
    agg.setdefault(k, [0, 0.0]); agg[k][0] += 1; agg[k][1] += us
for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f'  {k:10s} n={n:3d} {us/1e3:7.2f} ms')
print('--- conv/bn launches (time us, TFLOP/s, GB/s, bound us = max(flop/2.5P, bytes/6.3T))')
for ph, d, us, fl, by in rows:
    if fl or by:
        bound = max(fl / 2.5e15, by / 6.3e12) * 1e6
        print(f'{ph} {d:34s} {us:8.1f} us  {fl/us/1e6:7.1f} TF  {by/us/1e3:7.0f} GB/s  bound {bound:6.1f} us  eff {bound/us:5.2f}')
