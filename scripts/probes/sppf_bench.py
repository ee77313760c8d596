"""SPPF triple max-pool forward / backward alone on the yolov5s training shape (64 x 20 x 20 x 256 bf16) and the yolov5l one (x 512).
Usage: python scripts/probes/sppf_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hd_yolo_amd import ops, bench_util
dev = torch.device('cuda', 0)
for N, H, W, C in ((64, 20, 20, 256), (64, 20, 20, 512), (16, 40, 40, 256)):
    x = torch.randn((N, H, W, C), device=dev).to(torch.bfloat16)
    ys = [torch.empty_like(x) for _ in range(3)]
    idx = [torch.empty((N, H, W, C), dtype=torch.uint8, device=dev) for _ in range(3)]
    gs = [torch.randn((N, H, W, C), device=dev).to(torch.bfloat16) for _ in range(4)]
    dx = torch.empty_like(x)
    f = [ops.rec_sppf_pool_fwd(x, ys[0], ys[1], ys[2], idx)]
    b = [ops.rec_sppf_pool_bwd(gs[0], gs[1], gs[2], gs[3], idx, dx)]
    tf = bench_util.timed(lambda: ops.run(f), 20) * 1e3
    tb = bench_util.timed(lambda: ops.run(b), 20) * 1e3
    mb = x.numel() * 2 / 1e6
    print(f'sppf {N}x{H}x{W}x{C}: forward {tf:.1f} us ({mb * 4 + mb * 1.5:.0f} MB), backward {tb:.1f} us ({mb * 5 + mb * 1.5:.0f} MB)')
