"""SPPF pool kernels alone at the train-step shape (64 x 20 x 20 x 256 slices of a 1024-wide concat buffer) and how they scale with the batch (us).
Run: PYTHONPATH=. python scripts/probes/sppf_probe.py"""
import torch

from hd_yolo_amd import ops
from hd_yolo_amd.bench_util import time_record

DEV = 'cuda:0'
for N, H, C in [(64, 20, 256), (32, 20, 256), (16, 20, 256), (8, 20, 256), (64, 20, 128), (128, 32, 512)]:
    buf = torch.randn(N, H, H, 4 * C, device=DEV).bfloat16()
    sl = [buf[..., i * C:(i + 1) * C] for i in range(4)]
    idx = [torch.empty(N, H, H, C, dtype=torch.uint8, device=DEV) for _ in range(3)]
    a = time_record(ops.rec_sppf_pool_fwd(sl[0], sl[1], sl[2], sl[3], idx), reps=10)
    b = time_record(ops.rec_sppf_pool_fwd(sl[0], sl[1], sl[2], sl[3], None), reps=10)
    g = torch.randn(N, H, H, 4 * C, device=DEV).bfloat16()
    gs = [g[..., i * C:(i + 1) * C] for i in range(4)]
    dx = torch.empty(N, H, H, C, device=DEV).bfloat16()
    c = time_record(ops.rec_sppf_pool_bwd(gs[0], gs[1], gs[2], gs[3], idx, dx), reps=10) if H == 20 else float('nan')
    mb = N * H * H * C * 2 / 1e6
    print(f'N={N:3d} {H}x{H} C={C}: fwd with positions {a:6.1f}  without {b:6.1f}  bwd {c:6.1f} us   (plane tensor {mb:.1f} MB; HBM time of fwd ~{(4 * mb + 3 * mb / 2) / 6.3e3 * 1e3:.1f} us)', flush=True)
