"""Replay only the launch records of the C2 train plan whose layer_bench label matches a regex — the target of `rocprofv3 --pmc`
passes on single layers (per-dispatch counters of exactly these launches).
Usage: python scripts/layer_probe.py '<regex>' [reps=5] [variant=s] [batch=64] [size=640]"""
import os, re, sys
os.environ.setdefault('YOLOv5_VERBOSE', 'false')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hd_yolo_amd import synth, ops
from hd_yolo_amd.bench_util import describe, flat_records, time_record
from metayolo.models.yolo import Model

pat = re.compile(sys.argv[1])
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
variant = sys.argv[3] if len(sys.argv) > 3 else 's'
B = int(sys.argv[4]) if len(sys.argv) > 4 else 64
S = int(sys.argv[5]) if len(sys.argv) > 5 else 640
m = Model(synth.make_cfg(variant, 8), synth.make_hyp())
m.load_state_dict(synth.synth_state_dict(synth.shapes_of(m), seed=0), strict=False)
m = m.to('cuda:0').train(); m.half()
x = synth.synth_images(B, S, seed=0).to('cuda:0')
t = synth.synth_targets(B, S, 8, seed=1)
l, _ = m(x, t); l['det']['det_loss'].backward()
plan = next(iter(m._eng().plans.values()))
torch.cuda.synchronize()
seen = set()
# HDY_PROBE_MARKERS=1: a fill kernel on an int16 tensor of 65536 * (index + 1) elements is launched before every matched record, so that
# scripts/pmc_layers.py can cut a rocprofv3 counter CSV into per-layer segments (several layers run on the same kernel and grid)
markers = os.environ.get('HDY_PROBE_MARKERS') == '1'
mark = torch.empty(65536 * 64, dtype=torch.int16, device='cuda:0') if markers else None
for ph, recs in (('F', plan.fwd), ('B', plan.bwd)):
    for rec in flat_records(recs):
        d, fl, by = describe(rec)
        label = f'{ph} {d}'
        if not pat.search(label) or label in seen:
            continue
        seen.add(label)
        if markers:
            mark[:65536 * len(seen)].fill_(1)
        us = time_record(rec, reps)
        print(f'{label:44s} {us:8.1f} us  {fl/us/1e6:7.1f} TF  {by/us/1e3:7.0f} GB/s', flush=True)
