import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch, torch.nn.functional as F
from hd_yolo_amd import ops
import test_gpu_kernels as tk
DEV = tk.DEV
def run(case, dtype):
    N, H, W, C, K, R, stride, pad = case
    x = tk.q(tk.rnd((N, C, H, W), 1), dtype)
    w = tk.rnd((K, C, R, R), 2, (3.0 / (C * R * R)) ** 0.5)
    wq = tk.q(w, dtype)
    xd = tk.to_dev_nhwc(x, dtype, ld=C + 16, off=8)
    Ho, Wo = ops.out_dim(H, R, stride, pad), ops.out_dim(W, R, stride, pad)
    wp = ops.pack_alloc(K, C, R, R, stride, pad, ops.PACK_FWD, dtype, DEV)
    y = torch.zeros((N, Ho, Wo, K), dtype=dtype, device=DEV)
    ops.run([ops.rec_pack(w.to(DEV), None, stride, pad, ops.PACK_FWD, wp), ops.rec_conv_fwd(xd, wp, y, K, R, R, stride, pad)])
    ref = F.conv2d(x, wq, None, stride, pad)
    got = tk.from_dev_nhwc(y)
    err = (got - ref).abs()
    print(case, dtype, 'fwd max err', err.max().item(), 'ref max', ref.abs().max().item())
    bad = (err > 0.05 * ref.abs().max()).nonzero()
    print('  bad count', len(bad), 'of', err.numel())
    if len(bad):
        print('  bad n', sorted(set(bad[:, 0].tolist()))[:10], 'k', sorted(set(bad[:, 1].tolist()))[:40], 'h', sorted(set(bad[:, 2].tolist()))[:40], 'w', sorted(set(bad[:, 3].tolist()))[:40])
        # which input channels are missing: linear probe
    # channel probe: x = one-hot channel c, w = ones -> y = count of taps
    for c in range(0, C, max(1, C // 16)):
        xo = torch.zeros((N, C, H, W)); xo[:, c] = 1.0
        xod = tk.to_dev_nhwc(xo, dtype, ld=C + 16, off=8)
        wo = torch.ones((K, C, R, R))
        ops.run([ops.rec_pack(wo.to(DEV), None, stride, pad, ops.PACK_FWD, wp), ops.rec_conv_fwd(xod, wp, y, K, R, R, stride, pad)])
        g = tk.from_dev_nhwc(y); r = F.conv2d(xo, wo, None, stride, pad)
        e = (g - r).abs().max().item()
        if e > 0.01: print('   channel', c, 'err', e, 'got center', g[0, 0, Ho // 2, Wo // 2].item(), 'ref', r[0, 0, Ho // 2, Wo // 2].item())
for a in sys.argv[1:]:
    c, d = a.split(':')
    run(tk.CONV_CASES[int(c)], tk.DTYPES[int(d)])
