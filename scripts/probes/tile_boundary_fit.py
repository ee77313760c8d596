"""What does a tile boundary of conv_deep cost in the SHIPPED build (no stamps)?  1x1 convolutions over the same M x K output with C = 128 .. 1024
input channels: every workgroup walks the same number of tiles, only the K-tiles per tile change, so time = tiles_per_wg * (nkt * c_k + c_b) + c_0.
A least-squares line over nkt gives c_k (per K-tile) and the intercept (tiles_per_wg * c_b + c_0); two M give c_b and c_0 apart.
    HDY_LIB=<build>.so python scripts/probes/tile_boundary_fit.py [bn]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from hd_yolo_amd import _lib, ops
dev = torch.device('cuda', 0)
dt = torch.bfloat16
K = int(sys.argv[1]) if len(sys.argv) > 1 else 128
stats = len(sys.argv) > 2 and sys.argv[2] == 'stats'


def make(N, C):
    x = torch.randn((N, 64, 64, C), device=dev).to(dt)
    w = torch.randn((K, C, 1, 1), device=dev) * 0.05
    y = torch.empty((N, 64, 64, K), dtype=dt, device=dev)
    wp = ops.pack_alloc(K, C, 1, 1, 1, 0, ops.PACK_FWD, dt, dev)
    ops.run([ops.rec_pack(w, None, 1, 0, ops.PACK_FWD, wp)])
    st = torch.empty((ops.stat_slabs(N, 64, 64, C, K, 1, 1, 1, 0, dt), 2, K), dtype=torch.float32, device=dev) if stats else None
    return [ops.rec_conv_fwd(x, wp, y, K, 1, 1, 1, 0, stats=st)]


res = {}
for N in (32, 64):                       # 512 / 1024 row tiles of 256: 2 / 4 per workgroup at K = 128
    recs = {C: make(N, C) for C in (128, 256, 512, 1024)}
    for r in recs.values():
        for _ in range(3):
            ops.run(r)
    ts = {C: [] for C in recs}
    for rep in range(7):                 # interleaved rounds in one process
        for C, r in recs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.run(r)
            e1.record(); torch.cuda.synchronize()
            ts[C].append(e0.elapsed_time(e1) * 100)
    med = {C: float(np.median(v)) for C, v in ts.items()}
    nkt = np.array([C // 64 for C in med], dtype=float)
    t = np.array(list(med.values()))
    slope, icpt = np.polyfit(nkt, t, 1)
    res[N] = (slope, icpt)
    tiles = N * 4096 // 256 * (K // 128 if K >= 128 else 1) / 256.0
    print(f'N={N:3d} ({tiles:.0f} tiles per workgroup, {_lib.query("hdy_last_dispatch").decode()})  ' + '  '.join(f'C={C}: {v:7.1f} us' for C, v in med.items()) +
          f'   per K-tile {slope / tiles * 1000:6.0f} ns, intercept {icpt:6.1f} us', flush=True)
(s1, i1), (s2, i2) = res[32], res[64]
t1 = 32 * 16 * (K // 128) / 256.0
t2 = 2 * t1
cb = (i2 - i1) / (t2 - t1)
print(f'{os.environ.get("HDY_LIB", "libhdyolo_hip.so"):22s} K={K}{" stats" if stats else ""}: tile boundary {cb:6.2f} us, launch + prologue + tail {i1 - t1 * cb:6.2f} us, K-tile {s2 / t2 * 1000:5.0f} ns', flush=True)
