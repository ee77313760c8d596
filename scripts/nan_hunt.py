import os, sys
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
m = m.to(dev).train()
if sys.argv[1] == 'bf16': m.half()
opt = bench.make_optimizer(m, hyp, 64)
x = synth.synth_images(64, 640, seed=0).to(dev)
t = synth.synth_targets(64, 640, 8, seed=1)
for it in range(40):
    for j, g in enumerate(opt.param_groups):
        lo = hyp['warmup_bias_lr'] if j == 2 else 0.0
        g['lr'] = lo + (hyp['lr0'] * 1.0 - lo) * it / 100
        g['momentum'] = hyp['warmup_momentum'] + (hyp['momentum'] - hyp['warmup_momentum']) * it / 100
    l, _ = m(x, t); l['det']['det_loss'].backward()
    bad = [k for k, p in m.named_parameters() if not torch.isfinite(p.grad).all()]
    if bad or it % 8 == 0:
        plan = next(iter(m._eng().plans.values()))
        amax = max(float(v.t().float().abs().max()) for v in plan.vals if v.parts is None)
        print(it, round(l['det']['det_loss'].item(), 2), 'max |act|', amax, 'bad grads:', len(bad), bad[:6], bad[-3:], flush=True)
        wmax = max(float(p.abs().max()) for p in m.parameters())
        rv = min(float(b.min()) for k, b in m.named_buffers() if k.endswith('running_var'))
        print('   max |w|', wmax, 'min running_var', rv, flush=True)
    if bad: break
    opt.step(); opt.zero_grad(set_to_none=True)
