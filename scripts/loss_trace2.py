import os, sys
os.environ.setdefault('YOLOv5_VERBOSE', 'false')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hd_yolo_amd import synth
from metayolo.models.yolo import Model
import bench
dev = torch.device('cuda', 0)
hyp = synth.make_hyp()
for fused in ('1', '0'):
    os.environ['HDY_FUSED_LOSS'] = fused
    m = Model(synth.make_cfg('s', 8), hyp)
    m.load_state_dict(synth.synth_state_dict(synth.shapes_of(m), seed=0), strict=False)
    m = m.to(dev).train(); m.half()
    opt = bench.make_optimizer(m, hyp, 64)
    x = synth.synth_images(64, 640, seed=0).to(dev)
    t = synth.synth_targets(64, 640, 8, seed=1)
    out = []
    for it in range(26):
        for j, g in enumerate(opt.param_groups):
            lo = hyp['warmup_bias_lr'] if j == 2 else 0.0
            g['lr'] = lo + (hyp['lr0'] * 1.0 - lo) * it / 100
            g['momentum'] = hyp['warmup_momentum'] + (hyp['momentum'] - hyp['warmup_momentum']) * it / 100
        l, _ = m(x, t); l['det']['det_loss'].backward()
        gn = max(float(p.grad.abs().max()) for p in m.parameters())
        opt.step(); opt.zero_grad(set_to_none=True)
        out.append((round(l['det']['det_loss'].item(), 2), round(gn, 2)))
    print('fused' if fused == '1' else 'unfused', out, flush=True)
