"""Small host utilities (reference: metayolo/engines/general.py:116-128 init_seeds, :391 one_cycle, increment_path, colorstr)."""
import math
import os
import random
from pathlib import Path

import numpy as np
import torch


def init_seeds(seed=0):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def one_cycle(y1=0.0, y2=1.0, steps=100):
    """sinusoidal ramp from y1 to y2 over `steps` (https://arxiv.org/abs/1812.01187)"""
    return lambda x: ((1 - math.cos(x * math.pi / steps)) / 2) * (y2 - y1) + y1


def colorstr(*args):
    return str(args[-1])


def increment_path(path, exist_ok=False, sep='', mkdir=False):
    path = Path(path)
    if path.exists() and not exist_ok:
        base = path
        for n in range(2, 9999):
            path = Path(f'{base}{sep}{n}')
            if not path.exists():
                break
    if mkdir:
        path.mkdir(parents=True, exist_ok=True)
    return path


def print_args(name, opt):
    print(f'{name}: ' + ', '.join(f'{k}={v}' for k, v in vars(opt).items()))


# ------------------------------------------------------------------------------------------ checkpoint interop
def intersect_dicts(da, db, exclude=()):
    """Entries of da whose key is in db with the same shape, minus keys containing any `exclude` substring
    (reference: metayolo/engines/general.py:126-128)."""
    from collections import OrderedDict
    return OrderedDict((k, v) for k, v in da.items() if k in db and v.shape == db[k].shape and not any(x in k for x in exclude))


def checkpoint_state(ckpt, prefer_ema=False):
    """state_dict out of any checkpoint form: a bare state_dict, this build's {'model': state_dict, 'ema': state_dict}, or the
    reference's {'model': nn.Module.half(), 'ema': nn.Module.half()} (train.py:145-166, :530-540: loads
    `ckpt['model'].float().state_dict()`, EMA weights when restarting from a finished run)."""
    if not isinstance(ckpt, dict) or 'model' not in ckpt:
        m = ckpt
    else:
        m = ckpt['ema'] if prefer_ema and ckpt.get('ema') is not None else ckpt['model']
    if hasattr(m, "state_dict") and callable(m.state_dict):
        m = m.float().state_dict()
    return {k: (v.float() if hasattr(v, "is_floating_point") and v.is_floating_point() else v) for k, v in m.items()}


def convert_yolo_weights(model, weights):
    """Re-key a stock YOLOv5 state_dict (`model.<layer>.<rest>`, layers numbered through backbone, neck and Detect) to this
    package's `backbone.<i>` / `neck.<i - len(backbone)>` / `headers.<tag>.<rest>` keys (reference: engines/general.py:530-560).
    Layers after Detect (the upstream mask head: `m` -> `seg` with reversed level order, `header` -> `seg_h`) keep the reference's
    mapping so that a mask checkpoint converts to the same keys; everything else there is dropped."""
    from collections import OrderedDict
    nb, nn_ = len(model.backbone), len(model.neck)
    tag = next(iter(model.headers.keys()))          # single header assumed, as in the reference
    out = OrderedDict()
    for key, value in weights.items():
        parts = key.split('.')
        layer = int(parts[1])
        if layer < nb:
            parts[0] = 'backbone'
        elif layer < nb + nn_:
            parts[0], parts[1] = 'neck', str(layer - nb)
        elif layer == nb + nn_:
            parts[0], parts[1] = 'headers', tag
        else:
            parts[0], parts[1] = 'headers', tag
            parts[2] = {'m': 'seg', 'header': 'seg_h'}.get(parts[2])
            if parts[2] is None:
                continue
            if parts[2] == 'seg':
                parts[3] = str(3 - int(parts[3]))
        out['.'.join(parts)] = value
    return out


def load_weights_from_yolo(model, weights, exclude=('anchor',)):
    """convert_yolo_weights + shape-checked partial load (reference: engines/general.py:563-568)."""
    csd = intersect_dicts(convert_yolo_weights(model, weights), model.state_dict(), exclude=exclude)
    model.load_state_dict(csd, strict=False)
    return model


def manipulate_header_label_order(header, label_map, convert_masks=False):
    """Re-order / subset / extend the classes of a Detect header in place: new class i takes the weights of old class
    label_map[i] (0-based; -1 or out of range = a fresh, newly initialised class).  Box and objectness rows are carried over
    (reference: engines/general.py:572-604).  The header's launch plans are rebuilt on the next forward."""
    old_nc, old_no = header.nc, header.no
    header.nc = len(label_map)
    header.no = header.nc + 5
    new_rows, old_rows = [], []
    for a in range(header.na):
        new_rows += [a * header.no + j for j in range(5)]
        old_rows += [a * old_no + j for j in range(5)]
        for i, k in enumerate(label_map):
            if 0 <= k < old_nc:
                new_rows.append(a * header.no + 5 + i)
                old_rows.append(a * old_no + 5 + k)
    fresh = header.build_det_layers().to(next(header.m.parameters()).device)
    state = fresh.state_dict()
    for key, old in header.m.state_dict().items():
        state[key][new_rows] = old[old_rows].to(state[key].dtype)
    fresh.load_state_dict(state)
    header.m = fresh
    if convert_masks:
        header.mask_indices = header.mask_indices[[0] + [k + 1 for k in label_map]]
    return header
