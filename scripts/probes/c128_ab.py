"""A/B of the filter-resident 3x3 kernel for 128 input channels (conv3x3_c128.hip) against the deep-pipelined implicit GEMM on the same launch records:
train forward with BatchNorm sums, stride-1 data gradient (with and without accumulation), eval forward with SiLU; yolov5s B = 64 and yolov5l C4 sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hd_yolo_amd import _lib, ops
from hd_yolo_amd.bench_util import time_record
DEV = torch.device('cuda', 0)
dt = torch.bfloat16


def case(N, H, C, K, kind):
    R, pad = 3, 1
    x = torch.randn((N, H, H, C), device=DEV).to(dt)
    w = torch.randn((K, C, R, R), device=DEV) * (3.0 / (C * R * R)) ** 0.5
    y = torch.empty((N, H, H, K), dtype=dt, device=DEV)
    if kind.startswith('dgrad'):
        wp = ops.pack_alloc(K, C, R, R, 1, pad, ops.PACK_DGRAD, dt, DEV)
        ops.run([ops.rec_pack(w, None, 1, pad, ops.PACK_DGRAD, wp)])
        dx = torch.zeros((N, H, H, C), dtype=dt, device=DEV)
        return ops.rec_conv_dgrad(y.normal_(), wp, dx, R, R, 1, pad, accumulate=kind.endswith('acc'))
    wp = ops.pack_alloc(K, C, R, R, 1, pad, ops.PACK_FWD, dt, DEV)
    ops.run([ops.rec_pack(w, None, 1, pad, ops.PACK_FWD, wp)])
    if kind == 'train':
        st = torch.empty((ops.stat_slabs(N, H, H, C, K, R, R, 1, pad, dt), 2, K), dtype=torch.float32, device=DEV)
        return ops.rec_conv_fwd(x, wp, y, K, R, R, 1, pad, stats=st)
    sc, sh = torch.ones(K, device=DEV), torch.zeros(K, device=DEV)
    return ops.rec_conv_fwd(x, wp, y, K, R, R, 1, pad, scale=sc, shift=sh, act=ops.ACT_SILU)


for N, H, C, K, kind in [(64, 40, 128, 128, 'train'), (64, 40, 128, 128, 'dgrad'), (64, 40, 128, 128, 'dgrad_acc'), (128, 128, 128, 128, 'eval'), (16, 80, 128, 128, 'train'),
                         (32, 40, 128, 128, 'train')]:
    fl = 2.0 * N * H * H * K * C * 9
    res = {}
    for rnd in range(2):
        for off in (1, 0):
            with _lib.option('HDY_NO_CONV3X3_C128', off):
                rec = case(N, H, C, K, kind)
                _lib.dispatch_log(reset=True)
                us = time_record(rec, 10)
                res.setdefault(off, []).append((us, _lib.dispatch_log()[0]))
            del rec
            torch.cuda.empty_cache()
    a, b = min(u for u, _ in res[1]), min(u for u, _ in res[0])
    print(f'{kind:9s} N={N:3d} {H:3d}x{H:<3d} {C}->{K}: {res[1][0][1]:14s} {a:7.1f} us {fl / a / 1e6:7.1f} TF | {res[0][0][1]:12s} {b:7.1f} us {fl / b / 1e6:7.1f} TF = {fl / b / 1e6 / 2500:.3f} of peak  ratio {b / a:.3f}',
          flush=True)
