"""The per-step weight packing (one launch for all filters of the yolov5s train plan) alone.  Usage: python scripts/probes/pack_bench.py"""
import os, sys
os.environ.setdefault('YOLOv5_VERBOSE', 'false')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hd_yolo_amd import synth, bench_util
from metayolo.models.yolo import Model
V = sys.argv[1] if len(sys.argv) > 1 else 's'
BB = int(sys.argv[2]) if len(sys.argv) > 2 else 64
m = Model(synth.make_cfg(V, 8), synth.make_hyp())
m.load_state_dict(synth.synth_state_dict(synth.shapes_of(m), seed=0), strict=False)
m = m.to('cuda:0').train(); m.half()
x = synth.synth_images(BB, 640, seed=0).to('cuda:0')
t = synth.synth_targets(BB, 640, 8, seed=1)
for tt in t:
    for a in tt['anns']['det']:
        a['boxes'], a['labels'] = a['boxes'].cuda(), a['labels'].cuda()
l, _ = m(x, t); l['det']['det_loss'].backward()
plan = next(iter(m._eng().plans.values()))
ms = bench_util.timed(lambda: plan.packs.run(), 20)
nd = len(plan.packs.descs)
out = sum(d.rows_total * d.Kdp for d in plan.packs.descs)
print(f'pack: {nd} descriptors, {out / 1e6:.1f} M packed elements, {ms * 1e3:.1f} us per launch')
