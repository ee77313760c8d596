"""Building blocks of the detector with the reference's names, constructor signatures, attribute names and
state_dict keys (reference: metayolo/models/layers.py:18-41 Conv, :87-97 Bottleneck, :119-131 C3, :174-189 SPPF,
:264-271 Concat).  The modules hold parameters and describe structure; arithmetic happens in HIP plans
(hd_yolo_amd/plan.py).  Inside `Model` the whole backbone+neck is one plan; calling a block on its own runs a
one-module plan (forward only).  Blocks the n/s/m/l/l6 detection configs never instantiate (Ghost*, C3TR, Focus,
MixConv2d, ...) are out of scope and absent.
"""
import math  # noqa: F401  (kept for configs that eval expressions)
from typing import List

import torch
import torch.nn as nn

from ... import engine as _engine
from .activations import _get_activation_fn

__all__ = ['autopad', 'Conv', 'Bottleneck', 'C3', 'SPPF', 'Concat', 'nn', 'torch']


def autopad(k, p=None):
    """'same' padding for odd kernels when p is not given."""
    if p is not None:
        return p
    return k // 2 if isinstance(k, int) else [v // 2 for v in k]


class _HipBlock(nn.Module):
    """Common standalone forward: NCHW fp32 CUDA in, NCHW-shaped (channels-last strided) out."""

    def forward(self, x):
        return _engine.module_forward(self, x)


class Conv(_HipBlock):
    """conv2d (no bias) -> BatchNorm2d -> activation; after Model.fuse(): conv2d (bias) -> activation."""

    def __init__(self, in_channels, out_channels, kernel_size=1, stride=1, padding=None, groups=1, act=True):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, autopad(kernel_size, padding),
                              dilation=1, groups=groups, bias=False)
        self.bn = nn.BatchNorm2d(out_channels)
        self.act = _get_activation_fn(act)

    def forward_fuse(self, x):
        # same plan path: the planner sees that `bn` is gone and uses conv.bias in the epilogue
        return _engine.module_forward(self, x)


class Bottleneck(_HipBlock):
    """x + cv2(cv1(x)) when shortcut and c1 == c2, else cv2(cv1(x))."""

    def __init__(self, c1, c2, shortcut=True, g=1, e=0.5):
        super().__init__()
        hidden = int(c2 * e)
        self.cv1 = Conv(c1, hidden, 1, 1)
        self.cv2 = Conv(hidden, c2, 3, 1, groups=g)
        self.add = bool(shortcut and c1 == c2)


class C3(_HipBlock):
    """cv3(cat(m(cv1(x)), cv2(x))): CSP bottleneck with three convolutions."""

    def __init__(self, c1, c2, n=1, shortcut=True, g=1, e=0.5):
        super().__init__()
        hidden = int(c2 * e)
        self.cv1 = Conv(c1, hidden, 1, 1)
        self.cv2 = Conv(c1, hidden, 1, 1)
        self.cv3 = Conv(2 * hidden, c2, 1)
        self.m = nn.Sequential(*[Bottleneck(hidden, hidden, shortcut, g, e=1.0) for _ in range(n)])


class SPPF(_HipBlock):
    """cv2(cat(x', p(x'), p(p(x')), p(p(p(x'))))) with x' = cv1(x), p = maxpool k/1/k//2."""

    def __init__(self, c1, c2, k=5):
        super().__init__()
        hidden = c1 // 2
        self.cv1 = Conv(c1, hidden, 1, 1)
        self.cv2 = Conv(hidden * 4, c2, 1, 1)
        self.m = nn.MaxPool2d(kernel_size=k, stride=1, padding=k // 2)


class Concat(nn.Module):
    """Channel concat.  Inside a plan it is pure buffer placement; called directly it is torch.cat on whatever
    device the inputs are on (no arithmetic involved)."""

    def __init__(self, dimension=1):
        super().__init__()
        self.d = dimension

    def forward(self, x: List[torch.Tensor]):
        return torch.cat(x, self.d)
