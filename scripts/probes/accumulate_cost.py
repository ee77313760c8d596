"""Cost of accumulate=True (dx += ...) and of a residual in the conv epilogues at bench shapes: python scripts/probes/accumulate_cost.py"""
import torch
from hd_yolo_amd import ops, _lib
DEV = 'cuda:0'


def timed(rec, n=30):
    for _ in range(3): ops.run(rec)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): ops.run(rec)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def run(N, H, W, C, K, R):
    pad = R // 2
    dy = torch.randn((N, H, W, K), device=DEV).bfloat16()
    w = torch.randn((K, C, R, R), device=DEV) * 0.02
    wpd = ops.pack_alloc(K, C, R, R, 1, pad, ops.PACK_DGRAD, torch.bfloat16, DEV)
    dx = torch.zeros((N, H, W, C), dtype=torch.bfloat16, device=DEV)
    ops.run([ops.rec_pack(w, None, 1, pad, ops.PACK_DGRAD, wpd)])
    out = []
    for acc in (False, True):
        _lib.dispatch_log(reset=True)
        t = timed([ops.rec_conv_dgrad(dy, wpd, dx, R, R, 1, pad, accumulate=acc)])
        out.append('%s acc=%d %.1f us' % (_lib.dispatch_log()[0], acc, t))
    print('dgrad %dx%d C=%d<-K=%d k%d:' % (H, W, C, K, R), ' | '.join(out), ' (dx %.0f MB)' % (dx.numel() * 2 / 1e6), flush=True)


for c in [(64, 40, 40, 256, 256, 1), (64, 40, 40, 128, 128, 3), (64, 80, 80, 64, 64, 1), (64, 80, 80, 128, 128, 1), (64, 20, 20, 512, 512, 1), (64, 160, 160, 32, 32, 3), (64, 80, 80, 64, 64, 3)]:
    run(*c)
