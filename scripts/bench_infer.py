"""BASELINE.json configs[3] (C4): yolov5l, 8 classes, batch 128, 1024x1024 inference in bf16 — hd_yolo_amd.bench_util.infer_benchmark (the measurement
bench.py carries as `infer`: network + decode + NMS + outputs with the detection logits of the random-init network calibrated to unit spread and the
confidence threshold where a tile keeps ~1024 candidates, asserting that every tile keeps detections with finite boxes) plus NMS microseconds per tile on
the SURVEY §8d stress inputs.
Usage: python scripts/bench_infer.py [variant=l] [batch=128] [size=1024] [iters=5]"""
import os, sys, json
os.environ.setdefault('YOLOv5_VERBOSE', 'false')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hd_yolo_amd import synth, ops, bench_util

variant = sys.argv[1] if len(sys.argv) > 1 else 'l'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
S = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dev = torch.device('cuda', 0)
out = bench_util.infer_benchmark(variant, B, S, iters, dev)
out['mem_GB'] = round(torch.cuda.max_memory_allocated() / 2**30, 1)
# NMS stress sets: M survivors per tile
nms = {}
for M in (256, 1024, 4096, 16384):
    for max_det in (300, 2000):
        p = synth.synth_nms_preds(16, M, 8, size=S, seed=2).to(dev)
        ms = bench_util.timed(lambda: ops.nms_batched(p, 8, 0.15, 0.45, max_det), 5)
        nms[f'M={M},max_det={max_det}'] = round(ms / 16 * 1e3, 1)
out['nms_us_per_tile_batch16'] = nms
print(json.dumps(out))
