"""why does the C4-size eval keep no detections? (round 4 debugging aid)"""
import os, sys
os.environ.setdefault('YOLOv5_VERBOSE', 'false')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hd_yolo_amd import ops, synth
from metayolo.models.yolo import Model
DEV = 'cuda:0'
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
model = Model(synth.make_cfg('l', 8), synth.make_hyp())
model.load_state_dict(synth.synth_state_dict(synth.shapes_of(model), seed=0), strict=False)
model = model.to(DEV).eval().half()
x = synth.synth_images(B, 1024, seed=0).to(DEV)
head = model.headers['det']
if os.environ.get('CAL', '1') == '1':
    print('calibration factors', synth.calibrate_det_logits(model, x[:2].contiguous()))
with torch.no_grad():
    _, outs = model(x)
    plan = [pl for pl in model._eng().plans.values() if pl.det_views()[0].shape[0] == B][-1]
    print('plans', [tuple(pl.det_views()[0].shape) for pl in model._eng().plans.values()])
    for i, d in enumerate(plan.det_views()):
        print('level', i, tuple(d.shape), d.dtype, 'finite', bool(torch.isfinite(d).all()), 'min/max', float(d.min()), float(d.max()), 'obj logit mean', float(d[..., 4].mean()))
    preds = head.decode_all(plan.det_views())
    obj = preds[..., 4]
    print('preds', tuple(preds.shape), 'finite', bool(torch.isfinite(preds).all()), 'obj quantiles', [round(float(torch.quantile(obj[0], q)), 4) for q in (0.0, 0.5, 0.9, 0.99, 0.999, 1.0)])
    print('default conf', head.nms_params, 'n', [len(o['det']['boxes']) for o in outs][:8])
    k = float(torch.kthvalue(obj[0].float(), obj[0].numel() - 1024).values)
    print('kth', k, 'survivors tile0', int((obj[0] > k).sum()), 'wh min', float(preds[..., 2:4].min()), 'wh median', float(preds[..., 2:4].median()))
    head.nms_params = dict(head.nms_params, conf_thres=k)
    _, outs = model(x)
    print('calibrated n', [len(o['det']['boxes']) for o in outs][:8])
    res = ops.nms_batched(preds, head.nc, k, 0.45, 300)
    print('direct nms n_keep', res['n_keep'].tolist()[:8])
    small = ((preds[0, :, 2] >= 2) & (preds[0, :, 3] >= 2) & (obj[0] > k)).sum()
    print('survivors with w,h >= 2 in tile 0:', int(small))
