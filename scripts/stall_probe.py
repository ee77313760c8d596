import time, torch, sys
x = torch.zeros(1 << 20, device='cuda')
big = torch.zeros(1 << 28, device='cuda')   # 1 GiB
mode = sys.argv[1]
torch.cuda.synchronize()
def run(n, heavy):
    stamps = []
    t_prev = time.perf_counter()
    for i in range(n):
        if heavy:
            big.add_(1.0)      # ~0.4 ms of GPU work each: the queue stays full
        else:
            x.add_(1.0)
        if i % 50 == 49:
            t = time.perf_counter(); stamps.append(1e3 * (t - t_prev)); t_prev = t
    torch.cuda.synchronize()
    return stamps
for heavy in (False, True):
    s = run(6000 if not heavy else 1500, heavy)
    big_ones = [round(v, 1) for v in s if v > 8]
    print(f'heavy={heavy}: {len(s)} chunks of 50 launches, median {sorted(s)[len(s)//2]:.2f} ms, chunks > 8 ms: {big_ones[:20]}', flush=True)
