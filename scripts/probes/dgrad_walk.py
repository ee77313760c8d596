import torch, time
from hd_yolo_amd import ops, _lib
DEV = 'cuda:0'
def run(N, H, W, C, K):
    R, stride, pad = 3, 2, 1
    Ho, Wo = H // 2, W // 2
    dy = torch.randn((N, Ho, Wo, K), device=DEV).bfloat16()
    w = torch.randn((K, C, R, R), device=DEV) * 0.02
    wpd = ops.pack_alloc(K, C, R, R, stride, pad, ops.PACK_DGRAD, torch.bfloat16, DEV)
    dx = torch.zeros((N, H, W, C), dtype=torch.bfloat16, device=DEV)
    ops.run([ops.rec_pack(w, None, stride, pad, ops.PACK_DGRAD, wpd)])
    rec = [ops.rec_conv_dgrad(dy, wpd, dx, R, R, stride, pad)]
    out = []
    for off in (0, 1):
        with _lib.option('HDY_DEEP_WALK', 1 - off):
            _lib.dispatch_log(reset=True)
            for _ in range(3): ops.run(rec)
            name = _lib.dispatch_log()[0]
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): ops.run(rec)
            e1.record(); torch.cuda.synchronize()
            out.append((name, e0.elapsed_time(e1) / 50 * 1000))
    print((N, H, W, C, K), ' | '.join('%s %.1f us' % o for o in out), flush=True)
for c in [(64, 160, 160, 64, 128), (64, 80, 80, 128, 256), (64, 40, 40, 256, 512), (64, 80, 80, 128, 128), (64, 40, 40, 256, 256)]:      # the yolov5s bench shapes (B = 64)
    run(*c)
