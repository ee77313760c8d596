"""Ad-hoc: print the loss of the first steps of the bench workload."""
import os, sys
os.environ.setdefault('YOLOv5_VERBOSE', 'false')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hd_yolo_amd import synth
from metayolo.models.yolo import Model
B, S = int(sys.argv[1]), int(sys.argv[2])
for dt in ('f32', 'bf16'):
    m = Model(synth.make_cfg('s', 8), synth.make_hyp())
    m.load_state_dict(synth.synth_state_dict(synth.shapes_of(m), seed=0), strict=False)
    m = m.to('cuda:0').train()
    if dt == 'bf16':
        m.half()
    opt = torch.optim.SGD(m.parameters(), lr=float(sys.argv[3]), momentum=0.9, nesterov=True)
    x = synth.synth_images(B, S, seed=0).to('cuda:0')
    t = synth.synth_targets(B, S, 8, seed=1)
    out = []
    for i in range(8):
        l, _ = m(x, t)
        l['det']['det_loss'].backward()
        gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters())).item()
        opt.step(); opt.zero_grad(set_to_none=True)
        out.append((round(l['det']['det_loss'].item() / B, 4), round(gn, 3)))
    print(dt, out, flush=True)
