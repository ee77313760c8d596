"""Finalize inside the apply pass (hdy_bn_fin_act_fwd) against the two launches it replaces, replayed alone: microseconds per
[bump, finalize, apply] and per [bump, fused] at the shapes of the yolov5s train step, with the kernel's measurement variants
(HDY_DEEP_DEBUG bits: 1 first row requested after the wait, 2 long poll sleep, 4 no fence (wrong), 8 no wait (wrong)).
Needs scripts/probes/retired/bn_finalize_inside_apply.patch applied (git apply) and the library rebuilt.  Results: profiles/r05_finalize_inside_apply.txt"""
import torch

from hd_yolo_amd import _lib, ops

DEV = 'cuda:0'
SHAPES = [(64 * 160 * 160, 64, 800, 32), (64 * 160 * 160, 32, 800, None), (64 * 80 * 80, 128, 1000, 64), (64 * 80 * 80, 64, 512, None),
          (64 * 40 * 40, 256, 800, 128), (64 * 40 * 40, 128, 400, None), (64 * 20 * 20, 512, 400, 256), (64 * 20 * 20, 256, 200, None)]


def block_us(recs, reps=20, blocks=5):
    for _ in range(3):
        ops.run(recs)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(blocks + 1)]
    evs[0].record()
    for b in range(blocks):
        for _ in range(reps):
            ops.run(recs)
        evs[b + 1].record()
    torch.cuda.synchronize()
    return min(evs[b].elapsed_time(evs[b + 1]) for b in range(blocks)) / reps * 1e3


def main():
    print(f'{"M":>9} {"K":>4} {"slabs":>5} pair | bump+fin+apply  apply alone | fused: default  +1 late row  +2 long sleep  +4 no fence  +8 no wait')
    for M, K, mtiles, Ka in SHAPES:
        y = torch.randn(1, 1, M, K, device=DEV).bfloat16()
        stats = torch.rand(mtiles, 2, K, device=DEV) * 50
        stats[:, 1] += 400
        par = [torch.ones(K, device=DEV), torch.zeros(K, device=DEV), torch.zeros(K, device=DEV), torch.ones(K, device=DEV)]
        bn = lambda a, b: tuple(p[a:b] for p in par)
        coef = [torch.empty(K, device=DEV) for _ in range(4)]
        za = torch.empty(1, 1, M, Ka or K, dtype=torch.bfloat16, device=DEV)
        zb = torch.empty(1, 1, M, K - Ka, dtype=torch.bfloat16, device=DEV) if Ka else None
        sync = torch.zeros(2, dtype=torch.int32, device=DEV)
        flags = torch.zeros((K + 7) // 8, dtype=torch.int32, device=DEV)
        bump = ops.rec_sync_bump(sync)
        if Ka:
            sep = [bump, ops.rec_bn_finalize_pair(stats, mtiles, K, Ka, M, bn(0, Ka), bn(Ka, K), *coef), ops.rec_bn_act_fwd_pair(y, coef[0], coef[1], za, zb)]
        else:
            sep = [bump, ops.rec_bn_finalize(stats, mtiles, K, M, *bn(0, K), *coef), ops.rec_bn_act_fwd(y, coef[0], coef[1], za)]
        fused = [bump, ops.rec_bn_fin_act_fwd(stats, mtiles, M, bn(0, Ka or K), bn(Ka, K) if Ka else None, *coef, y, za, zb, flags, sync)]
        row = [block_us(sep), block_us([bump, sep[2]])]
        for var in (0, 1, 2, 4, 8):
            with _lib.option('HDY_DEEP_DEBUG', var):
                row.append(block_us(fused))
        print(f'{M:9d} {K:4d} {mtiles:5d} {str(Ka):>4} | ' + '  '.join(f'{v:8.1f}' for v in row) + f'   expired {int(sync[1])}', flush=True)


if __name__ == '__main__':
    main()
