import os, sys, time, gc
os.environ.setdefault('YOLOv5_VERBOSE', 'false')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hd_yolo_amd import synth
from metayolo.models.yolo import Model
import bench
dev = torch.device('cuda', 0)
hyp = synth.make_hyp()
m = Model(synth.make_cfg('s', 8), hyp)
m.load_state_dict(synth.synth_state_dict(synth.shapes_of(m), seed=0), strict=False)
m = m.to(dev).train(); m.half()
opt = bench.make_optimizer(m, hyp, 64)
for g in opt.param_groups: g['lr'] = 1e-5
x = synth.synth_images(64, 640, seed=0).to(dev)
t = synth.synth_targets(64, 640, 8, seed=1)
for tt in t:
    for a in tt['anns']['det']:
        a['boxes'], a['labels'] = a['boxes'].to(dev), a['labels'].to(dev)
NOOPT = len(sys.argv) > 1 and sys.argv[1] == 'noopt'
FOREACH_OFF = len(sys.argv) > 1 and sys.argv[1] == 'noforeach'
if FOREACH_OFF:
    for g in opt.param_groups: g['foreach'] = False
def step():
    l, _ = m(x, t); l['det']['det_loss'].backward()
    if not NOOPT:
        opt.step()
    opt.zero_grad(set_to_none=True)
for _ in range(5): step()
for mode in ('gc on',):
    if mode != 'gc on':
        gc.collect(); gc.freeze(); gc.disable()
    torch.cuda.synchronize()
    ts = []
    T0 = time.perf_counter()
    for i in range(20):
        t0 = time.perf_counter(); step()
        if 'sync' in mode: torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    torch.cuda.synchronize()
    tot = 1e3 * (time.perf_counter() - T0) / 20
    print(f'{mode}: avg {tot:.2f} ms/step; host per step: ' + ' '.join(f'{v:.0f}' for v in ts), flush=True)
