#!/usr/bin/env python3
"""Validation entry point with the reference's `run(model, dataloader, meta_info, callbacks, ...)` surface
(reference: val_nuclei.py:108-221), on the MI355X path: eval forward (HIP plan) -> decode + NMS kernels -> APMeter.

    python val_nuclei.py --variant s --nc 8 --imgsz 640 --batch-size 32 --batches 4
"""
import argparse
import os
import sys

import numpy as np
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')               # before the HIP runtime loads: see hd_yolo_amd/__init__.py
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from metayolo import LOGGER  # noqa: E402
from metayolo.engines.torch_utils import select_device, time_sync, to_device  # noqa: E402
from metayolo.models.metrics import APMeter  # noqa: E402


class _NoCallbacks:
    def run(self, *a, **k):
        pass


def flatten_onehot_objects(x):
    """(n, nc+1) multi-hot labels -> one row per (object, class) pair (reference val_nuclei.py:34-49)."""
    nbox, nc = x['labels'].shape
    keep = x['labels'].flatten() > 0.
    res = dict(x)
    res['labels'] = torch.tile(torch.arange(nc, device=x['labels'].device), (nbox,))[keep]
    res['labels'][res['labels'] == 0] = -100
    res['boxes'] = torch.repeat_interleave(x['boxes'], nc, 0)[keep]
    if 'scores' in x:
        res['scores'] = x['scores'].flatten()[keep]
    return res


def summarize_stats(ap_meter, task_id, **kwargs):
    """Precision / recall / F1 at the best mean-F1 operating point, AP@.5 and AP@.5:.95 per class; the summary averages the
    first four classes only, as the reference does for the NuCLS label set (val_nuclei.py:51-93)."""
    stats = ap_meter.ap_per_class(iouv=torch.linspace(0.5, 0.95, 10), ignore=[-100, -1])
    names = ap_meter.labels_text
    LOGGER.info(('%10s' * 2 + '%12s' * 5) % (task_id, 'Labels', 'P', 'R', 'F1', 'mAP@.5', 'mAP@.5:.95'))
    if not len(stats['labels']):
        return {'mp': 0.0, 'mr': 0.0, 'f1': 0.0, 'map50': 0.0, 'map': 0.0, 'fitness': 0.0}
    idx = stats['f1'].mean(0).argmax()
    p, r, f1 = stats['p'][:, idx], stats['r'][:, idx], stats['f1'][:, idx]
    ap50, ap = stats['ap'][:, 0], stats['ap'].mean(1)
    map50, map_ = float(ap50[:4].mean()), float(ap[:4].mean())
    mp, mr, mf1 = float(p[:4].mean()), float(r[:4].mean()), float(f1[:4].mean())
    pf = '%10s' + '%10i' + '%12.3g' * 5
    LOGGER.info(pf % ('all', sum(stats['counts']), mp, mr, mf1, map50, map_))
    for i, c in enumerate(stats['labels']):
        LOGGER.info(pf % (names.get(c, str(c)), stats['counts'][i], p[i], r[i], f1[i], ap50[i], ap[i]))
    return {'mp': mp, 'mr': mr, 'f1': mf1, 'map50': map50, 'map': map_, 'fitness': map50 * 0.1 + map_ * 0.9}


@torch.no_grad()
def run(model, dataloader, meta_info=None, callbacks=None, batch_size=32, half=True, verbose=False, save_txt=False,
        save_dir='', plots=False, epoch=0):
    callbacks = callbacks or _NoCallbacks()
    device = next(model.parameters()).device
    model.half() if half else model.float()
    model.eval()
    meters = {task_id: APMeter((meta_info or {}).get(task_id, {}).get('labels_text', {})) for task_id in model.headers}
    callbacks.run('on_val_start')
    dt, n_image = [0.0, 0.0, 0.0], 0
    for batch_i, (imgs, targets) in enumerate(dataloader):
        callbacks.run('on_val_batch_start')
        t1 = time_sync()
        imgs = torch.stack(list(imgs)).to(device, non_blocking=True).float()
        targets = to_device(targets, device)
        t2 = time_sync()
        dt[0] += t2 - t1
        _, outputs = model(imgs, compute_masks=False)
        t3 = time_sync()
        dt[1] += t3 - t2
        for output, target in zip(outputs, targets):
            n_image += 1
            for task_id in model.headers:
                o, t = output[task_id], dict(target['anns'][task_id][0])
                h, w = imgs.shape[-2:]
                if t['boxes'].numel() and float(t['boxes'].max()) <= 1.0:      # normalised training boxes -> pixels
                    t['boxes'] = t['boxes'] * t['boxes'].new_tensor([w, h, w, h])
                if o['labels'].dim() == 2:
                    o = flatten_onehot_objects(o)
                if t['labels'].dim() == 2:
                    t = flatten_onehot_objects(t)
                meters[task_id].add(o, t, iou_type='boxes')
        dt[2] += time_sync() - t3
        callbacks.run('on_val_batch_end')
    speeds = tuple(x / max(n_image, 1) * 1e3 for x in dt)
    val_stats = {task_id: summarize_stats(m, task_id=task_id) for task_id, m in meters.items()}
    fitness = sum(s['fitness'] for s in val_stats.values())
    model.float()
    return fitness, val_stats, speeds


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--variant', default='s')
    ap.add_argument('--nc', type=int, default=8)
    ap.add_argument('--imgsz', type=int, default=640)
    ap.add_argument('--batch-size', type=int, default=32)
    ap.add_argument('--batches', type=int, default=4)
    ap.add_argument('--weights', default='')
    ap.add_argument('--device', default='')
    ap.add_argument('--no-half', action='store_true')
    opt = ap.parse_args()
    from hd_yolo_amd import synth
    from metayolo.datasets import SyntheticTiles
    from metayolo.models.yolo import Model
    device = select_device(opt.device)
    model = Model(synth.make_cfg(opt.variant, opt.nc), synth.make_hyp())
    if opt.weights:
        from metayolo.engines.general import checkpoint_state, intersect_dicts
        ck = torch.load(opt.weights, map_location='cpu', weights_only=False)
        sd = checkpoint_state(ck, prefer_ema=True)        # this build's or the reference's checkpoint form; EMA weights as val.run gets them in train.py:487
        model.load_state_dict(intersect_dicts(sd, model.state_dict()), strict=False)
    else:
        model.load_state_dict(synth.synth_state_dict(synth.shapes_of(model), seed=0), strict=False)
    model = model.to(device)
    loader = SyntheticTiles(opt.batch_size, opt.imgsz, opt.nc, opt.batches, seed=12345)
    fitness, stats, speeds = run(model, loader, half=not opt.no_half)
    print(f'fitness {fitness:.4f}; ms/img pre {speeds[0]:.3f} infer+nms {speeds[1]:.3f} metrics {speeds[2]:.3f}')


if __name__ == '__main__':
    main()
