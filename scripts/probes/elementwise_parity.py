import os, sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
os.environ.setdefault('YOLOv5_VERBOSE', 'false')
from hd_yolo_amd import synth
from metayolo.models.yolo import Model
DEV = 'cuda:0'
for tag, variant in [('n_64', 'n'), ('s_128', 's'), ('n6_128', 'n6')]:
    g = np.load(f'tests/golden/stages_{tag}.npz')
    batch, size, nc = (int(v) for v in g['meta'])
    m = Model(synth.make_cfg(variant, nc), synth.make_hyp(conf_thres=float(g['conf_thres'])))
    m.load_state_dict(synth.synth_state_dict(synth.shapes_of(m), seed=0), strict=False)
    m = m.to(DEV).eval()
    x = synth.synth_images(batch, size, seed=7).to(DEV)
    with torch.no_grad():
        m(x)
        plan = next(iter(m._eng().plans.values()))
        for k in g.files:
            if k.startswith(('stage_', 'neck_', 'det_')):
                if k.startswith('det_'):
                    got = plan.det_views()[int(k[4:])]
                else:
                    got = plan.feature(int(k.split('_')[1]))
                got = torch.as_tensor(got).float().cpu().numpy(); ref = g[k].astype(np.float32)
                d = np.abs(got - ref); rms = np.sqrt((ref ** 2).mean()); mx = np.abs(ref).max()
                # smallest atol (in units of rms) with rtol 1e-4 that passes
                need = np.maximum(d - 1e-4 * np.abs(ref), 0).max() / rms
                print(f'{tag} {k:10s} max|d|/max {d.max()/mx:.1e}  max|d|/rms {d.max()/rms:.1e}  atol needed beside rtol 1e-4: {need:.1e} rms')
