"""Nearest buildable stand-in for BASELINE.json configs[4] (the reference's hnet is not runnable: SURVEY §8 f4): the metayolo
detector WITH its mask branch on 1280x1280 tiles, mixed det + mask loss, one train step timed.
Usage: python scripts/bench_mask.py [variant=s] [batch=16] [size=1280] [steps=5]"""
import os, sys, json
os.environ.setdefault('YOLOv5_VERBOSE', 'false')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hd_yolo_amd import synth
from metayolo.models.yolo import Model

variant = sys.argv[1] if len(sys.argv) > 1 else 's'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
S = int(sys.argv[3]) if len(sys.argv) > 3 else 1280
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dev = torch.device('cuda', 0)
cfg = synth.make_cfg(variant, 8)
cfg['headers'][0][3][3] = 1
m = Model(cfg, synth.make_hyp())
m.load_state_dict(synth.mask_state_dict(m), strict=False)
m = m.to(dev).train().half()
x = synth.synth_images(B, S, seed=0).to(dev)
targets = synth.synth_mask_targets(B, S, 8, per_image=40, seed=2)
opt = torch.optim.SGD(m.parameters(), lr=1e-4, momentum=0.9)


def step():
    losses, _ = m(x, targets, compute_masks=True)
    l = losses['det']
    (l['det_loss'] + l['mask_loss']).backward()
    opt.step()
    opt.zero_grad(set_to_none=True)            # as train.py does (torch's default)
    return l


for _ in range(2):
    l = step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(steps):
    l = step()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / steps
print(json.dumps({'config': f'yolov5{variant} + mask branch, nc=8, B={B}, {S}x{S}, bf16, det+mask loss', 'ms_per_step': round(ms, 2),
                  'tiles_per_s': round(B / ms * 1e3, 1), 'det_loss': float(l['det_loss']), 'mask_loss': float(l['mask_loss']),
                  'mem_GB': round(torch.cuda.max_memory_allocated() / 2**30, 1)}))
