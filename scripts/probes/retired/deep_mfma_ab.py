"""A/B of conv_deep.hip's 128-wide instance with v_mfma_f32_16x16x32_bf16 (HDY_DEEP_MFMA=16, shipped) against v_mfma_f32_32x32x16_bf16
(HDY_DEEP_MFMA=32): the same launch record timed alternately on one box (fastest of 5 blocks of 10, bench_util.time_record).
Rows: train forward with BatchNorm sums (raw epilogue), stride-1 data gradient, eval forward with SiLU; yolov5s B = 64 and yolov5l C4 shapes.
    python scripts/probes/deep_mfma_ab.py [quick]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hd_yolo_amd import _lib, ops
from hd_yolo_amd.bench_util import time_record
DEV = torch.device('cuda', 0)
dt = torch.bfloat16


def case(N, H, C, K, R, kind):
    pad = R // 2
    x = torch.randn((N, H, H, C), device=DEV).to(dt)
    w = torch.randn((K, C, R, R), device=DEV) * (3.0 / (C * R * R)) ** 0.5
    y = torch.empty((N, H, H, K), dtype=dt, device=DEV)
    if kind == 'dgrad':
        wp = ops.pack_alloc(K, C, R, R, 1, pad, ops.PACK_DGRAD, dt, DEV)
        ops.run([ops.rec_pack(w, None, 1, pad, ops.PACK_DGRAD, wp)])
        dx = torch.empty((N, H, H, C), dtype=dt, device=DEV)
        return ops.rec_conv_dgrad(y.normal_(), wp, dx, R, R, 1, pad), 2.0 * N * H * H * K * C * R * R
    wp = ops.pack_alloc(K, C, R, R, 1, pad, ops.PACK_FWD, dt, DEV)
    ops.run([ops.rec_pack(w, None, 1, pad, ops.PACK_FWD, wp)])
    if kind == 'train':
        slabs = ops.stat_slabs(N, H, H, C, K, R, R, 1, pad, dt)
        st = torch.empty((slabs, 2, K), dtype=torch.float32, device=DEV)
        return ops.rec_conv_fwd(x, wp, y, K, R, R, 1, pad, stats=st), 2.0 * N * H * H * K * C * R * R
    sc, sh = torch.ones(K, device=DEV), torch.zeros(K, device=DEV)
    return ops.rec_conv_fwd(x, wp, y, K, R, R, 1, pad, scale=sc, shift=sh, act=ops.ACT_SILU), 2.0 * N * H * H * K * C * R * R


CASES = [(64, 40, 128, 128, 3, 'train'), (64, 40, 128, 128, 3, 'dgrad'), (64, 20, 256, 256, 3, 'train'), (64, 20, 256, 256, 3, 'dgrad'),
         (64, 40, 256, 256, 1, 'train'), (64, 40, 128, 256, 1, 'dgrad'),
         (32, 40, 192, 192, 3, 'train'), (32, 40, 192, 192, 3, 'dgrad'), (32, 20, 384, 384, 3, 'train'),
         (128, 128, 128, 128, 3, 'eval'), (128, 64, 256, 256, 3, 'eval'), (128, 128, 128, 128, 1, 'eval'), (128, 32, 512, 512, 3, 'eval')]
if len(sys.argv) > 1 and sys.argv[1] == 'quick':
    CASES = CASES[:4]
for N, H, C, K, R, kind in CASES:
    res = {}
    for rnd in range(2):                       # alternating: 16, 32, 16, 32
        for mf in (16, 32):
            with _lib.option('HDY_DEEP_MFMA', mf), _lib.option('HDY_DEEP_BN', 128):
                rec, fl = case(N, H, C, K, R, kind)
                _lib.dispatch_log(reset=True)
                us = time_record(rec, 10)
                name = _lib.dispatch_log()[0]
            res.setdefault(mf, []).append((us, name))
            del rec
            torch.cuda.empty_cache()
    a, b = min(u for u, _ in res[16]), min(u for u, _ in res[32])
    print(f'{kind:5s} N={N:3d} {H:3d}x{H:<3d} {C:4d}->{K:4d} k{R}: 16x16x32 {a:7.1f} us {fl / a / 1e6:7.1f} TF [{res[16][0][1]}] | 32x32x16 {b:7.1f} us {fl / b / 1e6:7.1f} TF '
          f'[{res[32][0][1]}]  ratio {b / a:.3f}', flush=True)
