"""Ad-hoc: host enqueue time of the plan launch lists vs their GPU time."""
import os, sys, time
os.environ.setdefault('YOLOv5_VERBOSE', 'false')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hd_yolo_amd import synth, ops
from metayolo.models.yolo import Model
m = Model(synth.make_cfg('s', 8), synth.make_hyp())
m.load_state_dict(synth.synth_state_dict(synth.shapes_of(m), seed=0), strict=False)
m = m.to('cuda:0').train(); m.half()
x = synth.synth_images(64, 640, seed=0).to('cuda:0')
t = synth.synth_targets(64, 640, 8, seed=1)
for tt in t:
    for a in tt['anns']['det']:
        a['boxes'], a['labels'] = a['boxes'].cuda(), a['labels'].cuda()
opt = torch.optim.SGD(m.parameters(), lr=1e-4, momentum=0.9, nesterov=True)
def step():
    l, _ = m(x, t); l['det']['det_loss'].backward(); opt.step(); opt.zero_grad(set_to_none=True)
for _ in range(3): step()
plan = next(iter(m._eng().plans.values()))
torch.cuda.synchronize()
for name, recs in (('fwd', plan.fwd), ('bwd', plan.bwd)):
    t0 = time.perf_counter(); ops.run(recs); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f'{name}: {len(recs)} records, host enqueue {1e3*(t1-t0):.2f} ms, until done {1e3*(t2-t0):.2f} ms')
# phases of a step, host-side (no sync) and total
def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return r, 1e3 * (t1 - t0), 1e3 * (t2 - t0)
for _ in range(2):
    (l, _), h1, d1 = timed(lambda: m(x, t))
    _, h2, d2 = timed(lambda: l['det']['det_loss'].backward())
    _, h3, d3 = timed(lambda: (opt.step(), opt.zero_grad(set_to_none=True)))
    print(f'forward+loss host {h1:.2f} total {d1:.2f} | backward host {h2:.2f} total {d2:.2f} | opt host {h3:.2f} total {d3:.2f}')
