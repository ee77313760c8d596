"""The one stride-1 3x3 row at the 0.30 line: 256 -> 256 @20x20, B = 64, train forward with BatchNorm sums (and its data gradient) under the kernel
selection switches: deep pipeline (shipped), generic implicit GEMM (HDY_NO_DEEP), deep with fewer minimum tiles."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from contextlib import ExitStack
from hd_yolo_amd import _lib, ops
from hd_yolo_amd.bench_util import time_record
DEV = torch.device('cuda', 0)
dt = torch.bfloat16
N, H, C, K, R, pad = 64, 20, 256, 256, 3, 1
x = torch.randn((N, H, H, C), device=DEV).to(dt)
w = torch.randn((K, C, R, R), device=DEV) * 0.02
y = torch.empty((N, H, H, K), dtype=dt, device=DEV)
fl = 2.0 * N * H * H * K * C * 9
for name, opts in [('shipped', {}), ('generic igemm', {'HDY_NO_DEEP': 1}), ('no big igemm tiles', {'HDY_NO_DEEP': 1, 'HDY_NO_BIG_TILES': 1}), ('shipped again', {})]:
    with ExitStack() as es:
        for k, v in opts.items():
            es.enter_context(_lib.option(k, v))
        wp = ops.pack_alloc(K, C, R, R, 1, pad, ops.PACK_FWD, dt, DEV)
        ops.run([ops.rec_pack(w, None, 1, pad, ops.PACK_FWD, wp)])
        st = torch.empty((ops.stat_slabs(N, H, H, C, K, R, R, 1, pad, dt), 2, K), dtype=torch.float32, device=DEV)
        rec = ops.rec_conv_fwd(x, wp, y, K, R, R, 1, pad, stats=st)
        rec2 = ops.rec_conv_fwd(x, wp, y, K, R, R, 1, pad)
        _lib.dispatch_log(reset=True)
        a = min(time_record(rec, 10), time_record(rec, 10))
        b = min(time_record(rec2, 10), time_record(rec2, 10))
        print(f'{name:20s} with sums {a:6.1f} us = {fl / a / 1e6 / 2500:.3f} of peak | raw {b:6.1f} us = {fl / b / 1e6 / 2500:.3f}  [{_lib.dispatch_log()[0]}] slabs {st.shape[0]}', flush=True)
