"""Times the full-resolution kernels of the segmentation loss at the hnet bench size (16 x 1280 x 1280, 3 classes in 4-float pixels)."""
import torch
from hd_yolo_amd import ops
dev = 'cuda:0'
N, S, nc = 16, 1280, 3
low = torch.randn((N, S // 8, S // 8, 4), device=dev)
lab = torch.randint(0, nc, (N, S, S), device=dev)
masks = torch.nn.functional.one_hot(lab, nc).permute(0, 3, 1, 2).float().contiguous()
logits = ops.bilinear_fwd(low, (S, S))
up = torch.tensor([1.0], device=dev)


def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


mb = logits.numel() * 4 / 1e6
print('full-res logits %.0f MB, targets %.0f MB' % (mb, masks.numel() * 4 / 1e6))
print('clone of the logits      %7.1f us' % timed(lambda: logits.clone()))
print('bilinear_fwd fp32        %7.1f us' % timed(lambda: ops.bilinear_fwd(low, (S, S))))
print('softdice loss only       %7.1f us' % timed(lambda: ops.softdice(logits, masks, None)))
print('softdice loss + gradient %7.1f us' % timed(lambda: ops.softdice(logits, masks, None, upstream=up, want_grad=True)))
g = torch.randn_like(logits)
print('bilinear_bwd fp32        %7.1f us' % timed(lambda: ops.bilinear_bwd(g, (S // 8, S // 8))))
print('softdice_wgrad (fused)   %7.1f us' % timed(lambda: ops.softdice_wgrad(logits, masks, None, S // 8)))
dw = ops.softdice_wgrad(logits, masks, None, S // 8)[1]
dlow = torch.zeros((N, S // 8, S // 8, 8), device=dev)
print('H pass                   %7.1f us' % timed(lambda: ops.bilinear_bwd_h(dw, S // 8, dlow[..., :4])))
