import torch
x = torch.empty(1 << 28, dtype=torch.bfloat16, device='cuda')   # 512 MB
y = torch.empty_like(x)
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
ms = t(lambda: y.copy_(x)); print('copy 512MB->512MB', ms, 'ms', 2 * x.numel() * 2 / ms / 1e9, 'TB/s')
ms = t(lambda: torch.add(x, 1, out=y)); print('add', ms, 2 * x.numel() * 2 / ms / 1e9)
ms = t(lambda: x.sum()); print('sum (read only)', ms, x.numel() * 2 / ms / 1e9)
ms = t(lambda: y.fill_(1)); print('fill (write only)', ms, x.numel() * 2 / ms / 1e9)
z = torch.empty(1 << 26, dtype=torch.bfloat16, device='cuda'); w = torch.empty_like(z)
ms = t(lambda: w.copy_(z)); print('copy 128MB', ms, 2 * z.numel() * 2 / ms / 1e9)
