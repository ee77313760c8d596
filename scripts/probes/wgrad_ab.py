"""A/B of the weight-gradient kernel families on single shapes (bf16): the same launch record (kernel + split reduction) under process-wide options.
    python scripts/probes/wgrad_ab.py            # yolov5m shapes at B = 32 (round 6: the deep-pipelined kernel from K = 192, stride-1 3x3 on it)
Rows: N H C K R stride; columns: option sets.  Correctness of every variant against fp32 torch: tests/test_gpu_kernels.py CONV_CASES."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from contextlib import ExitStack
from hd_yolo_amd import _lib, ops
from hd_yolo_amd.bench_util import time_record
DEV = torch.device('cuda', 0)
dt = torch.bfloat16
SHAPES = [(32, 80, 192, 192, 3, 2), (32, 40, 192, 192, 3, 1), (32, 20, 384, 384, 3, 1), (32, 40, 384, 384, 3, 2), (32, 80, 192, 384, 3, 2), (32, 40, 384, 768, 3, 2),
          (64, 20, 256, 256, 3, 1), (64, 40, 128, 128, 3, 1), (64, 40, 256, 256, 3, 2)]
SETS = [('shipped r05', {'HDY_WGRAD_DEEP_KMIN': 256}), ('deep from K=192', {'HDY_WGRAD_DEEP_KMIN': 192})]     # (round 6 also measured the stride-1 3x3 layers on the deep kernel through a switch that is gone: profiles/r06_wgrad_deep_k192.txt)
for N, H, C, K, R, stride in SHAPES:
    pad = R // 2
    Ho = ops.out_dim(H, R, stride, pad)
    x = torch.randn((N, H, H, C), device=DEV).to(dt)
    dy = torch.randn((N, Ho, Ho, K), device=DEV).to(dt)
    g = torch.zeros((K, C, R, R), device=DEV)
    cells = []
    for name, opts in SETS:
        with ExitStack() as es:
            for k, v in opts.items():
                es.enter_context(_lib.option(k, v))
            wsb = _lib.query('hdy_conv_wgrad_workspace_bytes', N, H, H, C, K, R, R, stride, pad, ops.dcode(dt), 0)
            ws = torch.empty((wsb // 4 + 4,), dtype=torch.float32, device=DEV)
            rec = ops.rec_conv_wgrad(x, dy, g, None, R, R, stride, pad, ws)
            _lib.dispatch_log(reset=True)
            us = min(time_record(rec, 10), time_record(rec, 10))
            cells.append(f'{name}: {us:6.1f} us [{_lib.dispatch_log()[0]}]')
            del ws
    fl = 2.0 * N * Ho * Ho * K * C * R * R
    print(f'N={N} {H}x{H} {C}x{K} k{R} s{stride} ({fl / 1e9:.1f} GF): ' + ' | '.join(cells), flush=True)
