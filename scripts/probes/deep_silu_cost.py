"""What the SiLU of the eval-mode epilogue costs the deep-pipelined convolution: the same launch with act = SiLU and act = none (scale / shift only), us.
Run: PYTHONPATH=. python scripts/probes/deep_silu_cost.py"""
import torch

from hd_yolo_amd import _lib, ops
from hd_yolo_amd.bench_util import time_record

DEV = 'cuda:0'
for N, H, C, K, R in [(128, 64, 256, 256, 1), (128, 32, 512, 512, 1), (128, 128, 128, 128, 1), (128, 32, 1024, 512, 1), (128, 64, 512, 256, 1), (128, 64, 256, 256, 3),
                      (128, 128, 128, 128, 3), (128, 256, 64, 64, 3), (128, 256, 64, 64, 1)]:
    pad = R // 2
    x = torch.randn(N, H, H, C, device=DEV).bfloat16()
    w = torch.randn(K, C, R, R, device=DEV) * 0.05
    wp = ops.pack_alloc(K, C, R, R, 1, pad, ops.PACK_FWD, torch.bfloat16, DEV)
    ops.run([ops.rec_pack(w, None, 1, pad, ops.PACK_FWD, wp)])
    y = torch.empty(N, H, H, K, dtype=torch.bfloat16, device=DEV)
    sc, sh = torch.ones(K, device=DEV), torch.zeros(K, device=DEV)
    _lib.dispatch_log(reset=True)
    a = time_record(ops.rec_conv_fwd(x, wp, y, K, R, R, 1, pad, scale=sc, shift=sh, act=ops.ACT_SILU), reps=10)
    b = time_record(ops.rec_conv_fwd(x, wp, y, K, R, R, 1, pad, scale=sc, shift=sh, act=ops.ACT_NONE), reps=10)
    c = time_record(ops.rec_conv_fwd(x, wp, y, K, R, R, 1, pad, scale=sc, shift=sh, act=ops.ACT_RELU), reps=10)
    fl = 2.0 * N * H * H * K * C * R * R
    by = 2.0 * N * H * H * (C + K)
    print(f'{C:4d}->{K:4d} k{R} @{H}x{H} B={N}: SiLU {a:7.1f} us ({fl / a / 1e6:5.0f} TF, {by / a / 1e3:5.0f} GB/s)  none {b:7.1f}  ReLU {c:7.1f}  SiLU - none {a - b:6.1f}   {sorted(set(_lib.dispatch_log()))}', flush=True)
