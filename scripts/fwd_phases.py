import os, sys, time
os.environ.setdefault('YOLOv5_VERBOSE', 'false')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hd_yolo_amd import synth, ops, engine
from metayolo.models.yolo import Model
dev = torch.device('cuda', 0)
m = Model(synth.make_cfg('s', 8), synth.make_hyp())
m.load_state_dict(synth.synth_state_dict(synth.shapes_of(m), seed=0), strict=False)
m = m.to(dev).train(); m.half()
x = synth.synth_images(64, 640, seed=0).to(dev)
mode = sys.argv[1] if len(sys.argv) > 1 else 'a'
t = synth.synth_targets(64, 640, 8, seed=1)
for tt in t:
    for a in tt['anns']['det']:
        if mode == 'a':
            a['boxes'], a['labels'] = a['boxes'].to(dev), a['labels'].to(dev)
        else:
            a['boxes'], a['labels'] = a['boxes'].cuda(), a['labels'].cuda()
head = m.headers['det']
eng = m._eng()
def sync(): torch.cuda.synchronize(); return time.perf_counter()
for it in range(6):
    t0 = sync()
    plan = eng.plan_for(x, True, torch.bfloat16); plan.run_forward(x)
    t1 = sync()
    gts, lab = head.flatten_targets([tt['anns']['det'][0] for tt in t], dev)
    tcls = lab[:, 1:].float().contiguous(); gts = gts.contiguous()
    t2 = sync()
    plan.fused_loss(head)(gts, tcls)
    t3 = sync()
    loss = engine._FusedLossFn.apply(eng, plan, eng.hook, plan.loss_out[0:1])
    t4 = sync()
    loss.backward()
    t5 = sync()
    for p in m.parameters(): p.grad = None
    print(f'plan fwd {1e3*(t1-t0):.2f}  flatten {1e3*(t2-t1):.2f}  loss kernels {1e3*(t3-t2):.2f}  fn {1e3*(t4-t3):.2f}  bwd {1e3*(t5-t4):.2f}', flush=True)
print('--- now with bench.py optimizer and steps')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
hyp = synth.make_hyp()
opt = bench.make_optimizer(m, hyp, 64)
for g in opt.param_groups: g['lr'] = 1e-5
for it in range(8):
    t0 = sync()
    plan = eng.plan_for(x, True, torch.bfloat16); plan.run_forward(x)
    t1 = sync()
    gts, lab = head.flatten_targets([tt['anns']['det'][0] for tt in t], dev)
    tcls = lab[:, 1:].float().contiguous(); gts = gts.contiguous()
    t2 = sync()
    plan.fused_loss(head)(gts, tcls)
    t3 = sync()
    loss = engine._FusedLossFn.apply(eng, plan, eng.hook, plan.loss_out[0:1])
    loss.backward()
    t5 = sync()
    opt.step(); opt.zero_grad(set_to_none=True)
    t6 = sync()
    print(f'plan fwd {1e3*(t1-t0):.2f}  flatten {1e3*(t2-t1):.2f}  loss kernels {1e3*(t3-t2):.2f}  bwd {1e3*(t5-t3):.2f} opt {1e3*(t6-t5):.2f}', flush=True)
