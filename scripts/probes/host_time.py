"""host time per step of a bench script's `step()` against the GPU's: python scripts/probes/host_time.py scripts/bench_hnet.py s 16 1280 3"""
import sys, time, runpy
import torch
path = sys.argv[1]
sys.argv = sys.argv[1:]
g = runpy.run_path(path)
step = g['step']
for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for _ in range(n):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('host issue time per step %.2f ms; GPU drained %.2f ms after the last call; wall %.2f ms per step' % ((t1 - t0) / n * 1e3, (t2 - t1) * 1e3, (t2 - t0) / n * 1e3))
