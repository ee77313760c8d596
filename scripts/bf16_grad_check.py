"""Ad-hoc: how close are bf16-plan gradients to fp32-plan gradients, and how much of the gap is plain sensitivity
of this randomly-initialised BN network to a 2^-9 perturbation of its weights?"""
import os, sys
os.environ.setdefault('YOLOv5_VERBOSE', 'false')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hd_yolo_amd import synth
from metayolo.models.yolo import Model

def build(round_w=False):
    m = Model(synth.make_cfg('s', 8), synth.make_hyp())
    sd = synth.synth_state_dict(synth.shapes_of(m), seed=0)
    if round_w:
        sd = {k: (v.bfloat16().float() if k.endswith('conv.weight') or '.m.' in k and k.endswith('weight') else v) for k, v in sd.items()}
    m.load_state_dict(sd, strict=False)
    return m.to('cuda:0').train()

def grads(m, x, B, S):
    t = synth.synth_targets(B, S, 8, nmin=20, nmax=60, seed=5)
    l, _ = m(x, t)
    l['det']['det_loss'].backward()
    return l['det']['det_loss'].item(), {k: p.grad.clone() for k, p in m.named_parameters()}

def report(tag, a, b):
    cs = []
    for k in a[1]:
        u, v = a[1][k].flatten().double(), b[1][k].flatten().double()
        cs.append(((torch.dot(u, v) / (u.norm() * v.norm() + 1e-30)).item(), k))
    cs.sort()
    print(f'  {tag}: loss {a[0]:.5f} vs {b[0]:.5f}; worst cos {[(round(c, 3), k) for c, k in cs[:3]]} median {cs[len(cs)//2][0]:.4f}', flush=True)

for B, S in [(4, 256), (8, 640)]:
    x = synth.synth_images(B, S, seed=11).to('cuda:0')
    xr = x.bfloat16().float()
    print(f'B={B} S={S}')
    f32 = grads(build(), x, B, S)
    f32r = grads(build(round_w=True), xr, B, S)
    m = build(); m.half()
    b16 = grads(m, x, B, S)
    report('fp32 vs fp32 with bf16-rounded weights+input', f32, f32r)
    report('fp32 vs bf16 plan', f32, b16)
    report('fp32(rounded weights) vs bf16 plan', f32r, b16)
