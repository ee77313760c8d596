"""Where does the host wait?  Host-side duration of each phase of the C2 train step inside the free-running loop (no syncs added):
a phase that takes longer here than its pure enqueue cost is one in which the host blocks on the GPU."""
import os, sys, time
os.environ.setdefault('YOLOv5_VERBOSE', 'false')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from hd_yolo_amd import synth
from metayolo.models.yolo import Model

dev = torch.device('cuda', 0)
hyp = synth.make_hyp()
m = Model(synth.make_cfg('s', 8), hyp)
m.load_state_dict(synth.synth_state_dict(synth.shapes_of(m), seed=0), strict=False)
m = m.to(dev).train(); m.half()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
opt = bench.make_optimizer(m, hyp, B)
x = synth.synth_images(B, 640, seed=0).to(dev)
targets = synth.synth_targets(B, 640, 8, seed=1)
for t in targets:
    for a in t['anns']['det']:
        a['boxes'], a['labels'] = a['boxes'].to(dev), a['labels'].to(dev)
acc = [0.0] * 4
N = 60
per = []
evs = []
for it in range(N + 5):
    if it == 5:
        torch.cuda.synchronize(); acc = [0.0] * 4; T0 = time.perf_counter()
    ev = torch.cuda.Event(enable_timing=True); ev.record(); evs.append(ev)
    t0 = time.perf_counter()
    losses, _ = m(x, targets, compute_masks=False)
    t1 = time.perf_counter()
    losses['det']['det_loss'].backward()
    t2 = time.perf_counter()
    opt.step()
    t3 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    t4 = time.perf_counter()
    for i, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
        acc[i] += d
    per.append(round((t4 - t0) * 1e3, 1))
t_host = time.perf_counter() - T0
torch.cuda.synchronize()
t_all = time.perf_counter() - T0
print('host ms per step (forward+loss, backward, opt.step, zero_grad):', [round(a / N * 1e3, 3) for a in acc])
print('host ms of each step:', per[5:])
print('GPU ms between step starts after the sync:', [round(a.elapsed_time(b), 2) for a, b in zip(evs[5:35], evs[6:36])])
print('host loop %.3f ms/step, with final sync %.3f ms/step' % (t_host / N * 1e3, t_all / N * 1e3))
