"""Which torch-level ops (copies, fills, tiny tensor expressions) are still inside one bench train step: torch profiler, one step."""
import os, sys
os.environ.setdefault('YOLOv5_VERBOSE', 'false')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
from hd_yolo_amd import synth
from hd_yolo_amd.optim import SGD
from metayolo.models.yolo import Model

dev = torch.device('cuda', 0)
hyp = synth.make_hyp()
net = Model(synth.make_cfg('s', 8), hyp).to(dev).train()
x = synth.synth_images(64, 640, seed=0).to(dev)
targets = synth.synth_targets(64, 640, 8, seed=1)
opt = SGD(net.parameters(), lr=0.01, momentum=0.9, nesterov=True)


def step():
    losses, _ = net(x, targets, compute_masks=False)
    losses['det']['det_loss'].backward()
    opt.step()
    opt.zero_grad(set_to_none=True)


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
for r in sorted(prof.key_averages(), key=lambda r: -r.count):
    if r.key.startswith('aten::') or 'Memcpy' in r.key or 'Memset' in r.key or 'copyBuffer' in r.key or 'elementwise' in r.key:
        if r.count >= 1 and (r.device_time_total > 0 or r.key.startswith('aten::')):
            print('%-70s count %4d cpu %7.0f us dev %7.0f us' % (r.key[:70], r.count, r.cpu_time_total, r.device_time_total))
