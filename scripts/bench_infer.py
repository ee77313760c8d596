"""BASELINE.json configs[3] (C4): yolov5l, 8 classes, batch 128, 1024x1024 inference in bf16 — tiles/s of the eval launch list
(backbone + neck + det convs + decode + NMS) and NMS microseconds per tile on the SURVEY §8d stress inputs.
Usage: python scripts/bench_infer.py [variant=l] [batch=128] [size=1024] [iters=5]"""
import os, sys, json, time
os.environ.setdefault('YOLOv5_VERBOSE', 'false')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hd_yolo_amd import synth, ops
from metayolo.models.yolo import Model

variant = sys.argv[1] if len(sys.argv) > 1 else 'l'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
S = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dev = torch.device('cuda', 0)
m = Model(synth.make_cfg(variant, 8), synth.make_hyp())
m.load_state_dict(synth.synth_state_dict(synth.shapes_of(m), seed=0), strict=False)
m = m.to(dev).eval().half()
x = synth.synth_images(B, S, seed=0).to(dev)


def timed(fn, n):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


out = {'config': f'yolov5{variant} nc=8 B={B} {S}x{S} bf16 eval', 'mem_GB': None}
with torch.no_grad():
    ms_all = timed(lambda: m(x), iters)
    eng = m._eng()
    plan = next(iter(eng.plans.values()))
    ms_net = timed(lambda: plan.run_forward(x), iters)
    head = m.headers['det']
    dets = plan.det_views()
    ms_dec = timed(lambda: head.decode_all(dets), iters)
    preds = head.decode_all(dets)
    ms_out = timed(lambda: head.compute_outputs(preds), iters)
    conf, iou, max_det = head.nms_params['conf_thres'], head.nms_params['iou_thres'], int(head.nms_params['max_det'])
    ms_nms = timed(lambda: ops.nms_batched(preds, head.nc, conf, iou, max_det), iters)
ncand = preds.shape[1]
dec_bytes = B * ncand * (head.no + head.no + 1) * 4            # logits read + rows written (108 B per candidate at nc = 8)
out.update(ms_per_batch=round(ms_all, 3), tiles_per_s=round(B / ms_all * 1e3, 1), ms_network=round(ms_net, 3), ms_decode=round(ms_dec, 3),
           decode_us_per_tile=round(ms_dec / B * 1e3, 2), decode_GBs=round(dec_bytes / ms_dec / 1e6, 1), decode_hbm_frac=round(dec_bytes / ms_dec / 1e6 / 8000, 3),
           ms_nms_kernel=round(ms_nms, 3), nms_us_per_tile=round(ms_nms / B * 1e3, 2), ms_outputs_total=round(ms_out, 3),
           candidates_per_tile=ncand, mem_GB=round(torch.cuda.max_memory_allocated() / 2**30, 1))
gf = {'n': 4.13, 's': 15.81, 'm': 47.94, 'l': 107.76}[variant] * (S / 640) ** 2
out['network_TFLOPs'] = round(gf * B / ms_net, 1)
# NMS stress sets: M survivors per tile
nms = {}
for M in (256, 1024, 4096, 16384):
    for max_det in (300, 2000):
        p = synth.synth_nms_preds(16, M, 8, size=S, seed=2).to(dev)
        ms = timed(lambda: ops.nms_batched(p, 8, 0.15, 0.45, max_det), 5)
        nms[f'M={M},max_det={max_det}'] = round(ms / 16 * 1e3, 1)
out['nms_us_per_tile_batch16'] = nms
print(json.dumps(out))
