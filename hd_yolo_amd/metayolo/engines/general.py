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
