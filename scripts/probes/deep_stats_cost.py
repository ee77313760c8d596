"""What the BatchNorm sums cost the deep-pipelined forward: the same launch with and without the statistics slabs, replayed alone (us).
Run: PYTHONPATH=. python scripts/probes/deep_stats_cost.py"""
import torch

from hd_yolo_amd import _lib, ops
from hd_yolo_amd.bench_util import time_record

DEV = 'cuda:0'
for N, H, C, K, R in [(64, 40, 128, 128, 3), (64, 40, 128, 128, 1), (64, 20, 256, 256, 3), (64, 20, 512, 512, 1), (64, 20, 256, 256, 1), (64, 40, 256, 256, 1), (64, 80, 128, 128, 1)]:
    pad = R // 2
    x = torch.randn(N, H, H, C, device=DEV).bfloat16()
    w = torch.randn(K, C, R, R, device=DEV) * 0.05
    wp = ops.pack_alloc(K, C, R, R, 1, pad, ops.PACK_FWD, torch.bfloat16, DEV)
    ops.run([ops.rec_pack(w, None, 1, pad, ops.PACK_FWD, wp)])
    y = torch.empty(N, H, H, K, dtype=torch.bfloat16, device=DEV)
    ns = ops.stat_slabs(N, H, H, C, K, R, R, 1, pad, torch.bfloat16)
    stats = torch.empty(ns, 2, K, device=DEV)
    _lib.dispatch_log(reset=True)
    a = time_record(ops.rec_conv_fwd(x, wp, y, K, R, R, 1, pad, stats=stats), reps=20)
    b = time_record(ops.rec_conv_fwd(x, wp, y, K, R, R, 1, pad), reps=20)
    print(f'{C:4d}->{K:4d} k{R} @{H}x{H} slabs {ns:4d}: with sums {a:6.1f}  without {b:6.1f}  diff {a - b:5.1f} us   {sorted(set(_lib.dispatch_log()))}', flush=True)
