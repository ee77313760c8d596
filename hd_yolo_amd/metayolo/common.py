"""Host-side helpers the training entry point uses (reference: metayolo/common.py:41-47 is_parallel / de_parallel,
:128-159 ModelEMA)."""
import math
from copy import deepcopy

import torch
import torch.nn as nn


def is_parallel(model):
    return type(model) in (nn.parallel.DataParallel, nn.parallel.DistributedDataParallel) or hasattr(model, 'hdy_dp_module')


def de_parallel(model):
    if hasattr(model, 'hdy_dp_module'):
        return model.hdy_dp_module
    return model.module if is_parallel(model) else model


class ModelEMA:
    """Exponential moving average of the whole state_dict, decay 0.9999 * (1 - exp(-updates / 2000))."""

    def __init__(self, model, decay=0.9999, updates=0):
        src = de_parallel(model)
        eng = src.__dict__.pop('_hdy_engine', None)       # plans hold device buffers: never deep-copy them
        self.ema = deepcopy(src).eval()
        if eng is not None:
            object.__setattr__(src, '_hdy_engine', eng)
        self.updates = updates
        self.decay = lambda x: decay * (1 - math.exp(-x / 2000))
        for p in self.ema.parameters():
            p.requires_grad_(False)

    def update(self, model):
        with torch.no_grad():
            self.updates += 1
            d = self.decay(self.updates)
            msd = de_parallel(model).state_dict()
            fl = [(v, msd[k].detach()) for k, v in self.ema.state_dict().items() if v.dtype.is_floating_point]
            dst, src = [a for a, _ in fl], [b for _, b in fl]
            torch._foreach_mul_(dst, d)
            torch._foreach_add_(dst, src, alpha=1 - d)

    def update_attr(self, model, include=(), exclude=('process_group', 'reducer')):
        for k, v in model.__dict__.items():
            if (include and k not in include) or k.startswith('_') or k in exclude:
                continue
            setattr(self.ema, k, v)
