"""Small-batch inference latency of yolov5s (bf16, 640x640): network only and end to end (decode + NMS + outputs), per batch size.
Usage: python scripts/bench_latency.py [variant=s] [size=640]"""
import os, sys, json, time
os.environ.setdefault('YOLOv5_VERBOSE', 'false')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hd_yolo_amd import synth
from metayolo.models.yolo import Model

variant = sys.argv[1] if len(sys.argv) > 1 else 's'
S = int(sys.argv[2]) if len(sys.argv) > 2 else 640
dev = torch.device('cuda', 0)
m = Model(synth.make_cfg(variant, 8), synth.make_hyp(conf_thres=0.25))
m.load_state_dict(synth.synth_state_dict(synth.shapes_of(m), seed=0), strict=False)
m = m.to(dev).eval().half()
out = {}
with torch.no_grad():
    for B in (1, 4, 16, 64):
        x = synth.synth_images(B, S, seed=B).to(dev)
        for _ in range(5):
            m(x)
        plan = m._eng().plan_for(x, False, torch.bfloat16)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            plan.run_forward(x)
        torch.cuda.synchronize()
        net = (time.perf_counter() - t0) / 20 * 1e3
        t0 = time.perf_counter()
        for _ in range(20):
            m(x)
        torch.cuda.synchronize()
        full = (time.perf_counter() - t0) / 20 * 1e3
        out[f'B={B}'] = {'network_ms': round(net, 3), 'end_to_end_ms': round(full, 3), 'tiles_per_s': round(B / full * 1e3, 1)}
print(json.dumps({'config': f'yolov5{variant} nc=8 {S}x{S} bf16 eval, graph={os.environ.get("HDY_GRAPH", "0")}', **out}))
