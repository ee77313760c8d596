"""What the residual operand costs the eval-mode convolution (scale / shift + SiLU epilogue): the same launch with and without `res`, replayed alone (us).
Run: PYTHONPATH=. python scripts/probes/deep_res_cost.py"""
import torch

from hd_yolo_amd import _lib, ops
from hd_yolo_amd.bench_util import time_record

DEV = 'cuda:0'
for N, H, C, K, R in [(128, 128, 128, 128, 3), (128, 64, 256, 256, 3), (128, 32, 512, 512, 3), (128, 256, 64, 64, 3), (64, 40, 128, 128, 3)]:
    pad = R // 2
    x = torch.randn(N, H, H, C, device=DEV).bfloat16()
    w = torch.randn(K, C, R, R, device=DEV) * 0.05
    wp = ops.pack_alloc(K, C, R, R, 1, pad, ops.PACK_FWD, torch.bfloat16, DEV)
    ops.run([ops.rec_pack(w, None, 1, pad, ops.PACK_FWD, wp)])
    y = torch.empty(N, H, H, K, dtype=torch.bfloat16, device=DEV)
    res = torch.randn(N, H, H, K, device=DEV).bfloat16()
    sc, sh = torch.ones(K, device=DEV), torch.zeros(K, device=DEV)
    _lib.dispatch_log(reset=True)
    a = time_record(ops.rec_conv_fwd(x, wp, y, K, R, R, 1, pad, scale=sc, shift=sh, act=ops.ACT_SILU, res=res), reps=10)
    b = time_record(ops.rec_conv_fwd(x, wp, y, K, R, R, 1, pad, scale=sc, shift=sh, act=ops.ACT_SILU), reps=10)
    fl = 2.0 * N * H * H * K * C * R * R
    print(f'{C:4d}->{K:4d} k{R} @{H}x{H} B={N}: with res {a:7.1f} us ({fl / a / 1e6:6.0f} TF)  without {b:7.1f} us ({fl / b / 1e6:6.0f} TF)  diff {a - b:6.1f}   {sorted(set(_lib.dispatch_log()))}', flush=True)
