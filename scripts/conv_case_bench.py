"""Time one conv forward (bf16, scale/shift + SiLU epilogue) through the C ABI: python scripts/conv_case_bench.py N H W C K R.
With HDY_LIB=<other .so under csrc/build/> the same case runs on another build of the library (kernel A/B on one box)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hd_yolo_amd import _lib, ops
dev = torch.device('cuda', 0)
N, H, W, C, K, R = [int(v) for v in (sys.argv[1:7] if len(sys.argv) > 6 else (128, 64, 64, 256, 256, 3))]
dt = torch.bfloat16
x = torch.randn((N, H, W, C), device=dev).to(dt)
w = torch.randn((K, C, R, R), device=dev) * 0.05
y = torch.empty((N, H, W, K), dtype=dt, device=dev)
wp = ops.pack_alloc(K, C, R, R, 1, R // 2, ops.PACK_FWD, dt, dev)
sc = torch.ones(K, device=dev); sh = torch.zeros(K, device=dev)
ops.run([ops.rec_pack(w, None, 1, R // 2, ops.PACK_FWD, wp)])
if os.environ.get('ACT', '1') == '0':      # raw output (train forward / data gradient epilogue)
    rec = [ops.rec_conv_fwd(x, wp, y, K, R, R, 1, R // 2)]
else:
    rec = [ops.rec_conv_fwd(x, wp, y, K, R, R, 1, R // 2, scale=sc, shift=sh, act=ops.ACT_SILU)]
for _ in range(3): ops.run(rec)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): ops.run(rec)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 100
print(f'{us:.1f} us  {2.0*N*H*W*K*C*R*R/us/1e6:.1f} TF  [{_lib.query("hdy_last_dispatch").decode()}] act={os.environ.get("ACT", "1")}')
