import sys, runpy, cProfile, pstats, io, torch
sys.argv = ['scripts/bench_hnet.py', 's', '16', '1280', '2']
g = runpy.run_path('scripts/bench_hnet.py')
step = g['step']
for _ in range(3): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45); print(s.getvalue()[:9000])
