"""BASELINE.json configs[4] per-GPU shape: hnet multi-level (conv backbone + pyramid shared by a detection header and a PanopticSeg
semantic-segmentation header), 1280x1280 tiles, mixed det + seg loss, one training step timed (bf16).
Usage: python scripts/bench_hnet.py [variant=s] [batch=16] [size=1280] [steps=5] [seg_classes=3]"""
import os, sys, json
os.environ.setdefault('YOLOv5_VERBOSE', 'false')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hd_yolo_amd import synth
from hnet.hnet import HNet

variant = sys.argv[1] if len(sys.argv) > 1 else 's'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
S = int(sys.argv[3]) if len(sys.argv) > 3 else 1280
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
ncls = int(sys.argv[5]) if len(sys.argv) > 5 else 3
dev = torch.device('cuda', 0)
cfg = {'backbone': {'type': 'yolov5', 'cfg': synth.make_cfg(variant, 8), 'hyp': synth.make_hyp()},
       'headers': {'seg': {'type': 'PanopticSeg', 'configs': {'num_classes': ncls, 'feature_maps': None, 'in_channels': None, 'scale_factor': 8,
                                                               'resize_mode': 'bilinear', 'class_weight': None, 'roi_size': None}}}}
m = HNet(cfg)
m.detector.load_state_dict(synth.synth_state_dict(synth.shapes_of(m.detector), seed=0), strict=False)
m = m.to(dev).train().half()
x = synth.synth_images(B, S, seed=0).to(dev)
det_t = synth.synth_targets(B, S, 8, seed=1)
g = torch.Generator().manual_seed(3)
lab = torch.randint(0, ncls, (B, S // 16, S // 16), generator=g)                      # blocky synthetic tissue regions
lab = lab.repeat_interleave(16, 1).repeat_interleave(16, 2).to(dev)
masks = torch.nn.functional.one_hot(lab, ncls).permute(0, 3, 1, 2).float().contiguous()
targets = []
for i, t in enumerate(det_t):
    anns = {k: [{kk: (vv.to(dev) if torch.is_tensor(vv) else vv) for kk, vv in a.items()} for a in v] for k, v in t['anns'].items()}
    anns['seg'] = [{'roi': torch.tensor([0.0, 0.0, S, S]), 'masks': masks[i]}]
    targets.append({**t, 'anns': anns})
from hd_yolo_amd.optim import SGD          # torch.optim.SGD's update in one launch over all tensors (csrc/optim.hip), as bench.py / train.py use it
opt = SGD(m.parameters(), lr=1e-4, momentum=0.9)


def step():
    losses, _ = m(x, targets)
    (losses['det_det_loss'] + losses['seg_soft_iou_loss']).backward()
    opt.step()
    opt.zero_grad(set_to_none=True)            # as train.py does (and torch's default): the engine hands out fresh views of its flat gradient buffer
    return losses


for _ in range(2):
    l = step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(steps):
    l = step()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / steps
print(json.dumps({'config': f'hnet: yolov5{variant} backbone + pyramid, Detect (nc=8) + PanopticSeg ({ncls} classes, full-resolution masks), B={B}, '
                            f'{S}x{S}, bf16, det + seg loss', 'ms_per_step': round(ms, 2), 'tiles_per_s': round(B / ms * 1e3, 1),
                  'det_loss': float(l['det_det_loss']), 'seg_loss': float(l['seg_soft_iou_loss']),
                  'mem_GB': round(torch.cuda.max_memory_allocated() / 2**30, 1)}))
