"""Time hdy_conv_wgrad on one 3x3 layer shape (for rocprofv3 --kernel-trace --stats: splits the main kernel from the slab reduce).
Usage: python scripts/wgrad_probe.py N H W C K [reps]"""
import os, sys
os.environ.setdefault('YOLOv5_VERBOSE', 'false')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hd_yolo_amd import ops
N, H, W, C, K = (int(v) for v in sys.argv[1:6])
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 20
dt = torch.bfloat16
x = torch.randn((N, H, W, C), device='cuda').to(dt)
dy = torch.randn((N, H, W, K), device='cuda').to(dt)
g = torch.zeros((K, C, 3, 3), device='cuda')
ws = torch.empty(ops.wgrad_ws_bytes(N, H, W, C, K, 3, 3, 1, 1, dt) // 4 + 16, dtype=torch.float32, device='cuda')
rec = [ops.rec_conv_wgrad(x, dy, g, None, 3, 3, 1, 1, ws)]
for _ in range(3):
    ops.run(rec)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    ops.run(rec)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
print(f'wgrad {C}x{K} k3 @{H}x{W} N={N}: {us:.1f} us, {2.0*N*H*W*C*K*9/us/1e6:.0f} TFLOP/s, workspace {ws.numel()*4/2**20:.1f} MiB')
