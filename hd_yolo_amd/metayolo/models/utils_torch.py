"""Model utilities with the reference's names (reference: metayolo/models/utils_torch.py:37-51, :79-124, :140-178, :232-238).
Host-side only: parameter bookkeeping, BN folding (a few small tensor expressions at deploy time), logging."""
import math
from copy import deepcopy  # noqa: F401

import torch
import torch.nn as nn

from .. import LOGGER, check_version  # noqa: F401


def torch_meshgrid(*tensors):
    return torch.meshgrid(*tensors, indexing='ij')


def intersect_dicts(da, db, exclude=()):
    """Entries of da whose key is in db with the same shape and contains none of the `exclude` substrings."""
    return {k: v for k, v in da.items() if k in db and v.shape == db[k].shape and not any(x in k for x in exclude)}


def initialize_weights(model):
    """BatchNorm eps / momentum and in-place activations exactly as the reference sets them (:42-51)."""
    for m in model.modules():
        if type(m) is nn.BatchNorm2d:
            m.eps, m.momentum = 1e-3, 0.03
        elif type(m) in (nn.Hardswish, nn.LeakyReLU, nn.ReLU, nn.ReLU6, nn.SiLU):
            m.inplace = True


def fuse_conv_and_bn(conv, bn):
    """W' = diag(gamma / sqrt(var + eps)) W,  b' = beta - gamma * mean / sqrt(var + eps) (+ scaled conv bias)."""
    fused = nn.Conv2d(conv.in_channels, conv.out_channels, kernel_size=conv.kernel_size, stride=conv.stride,
                      padding=conv.padding, groups=conv.groups, bias=True).requires_grad_(False).to(conv.weight.device)
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    with torch.no_grad():
        fused.weight.copy_(conv.weight * scale.view(-1, 1, 1, 1))
        b = conv.bias if conv.bias is not None else torch.zeros_like(bn.running_mean)
        fused.bias.copy_((b - bn.running_mean) * scale + bn.bias)
    return fused


def model_info(model, verbose=False, img_size=640):
    n_p = sum(p.numel() for p in model.parameters())
    n_g = sum(p.numel() for p in model.parameters() if p.requires_grad)
    if verbose:
        print(f"{'layer':>5} {'name':>40} {'gradient':>9} {'parameters':>12} {'shape':>20} {'mu':>10} {'sigma':>10}")
        for i, (name, p) in enumerate(model.named_parameters()):
            print('%5g %40s %9s %12g %20s %10.3g %10.3g' % (i, name, p.requires_grad, p.numel(), list(p.shape), p.mean(), p.std()))
    LOGGER.info(f'Model summary: {len(list(model.modules()))} layers, {n_p} parameters, {n_g} gradients')


def scale_img(img, ratio=1.0, same_shape=False, gs=32):
    """Test-time-augmentation helper (reference: utils_torch.py:127-137): bilinear resize of a (bs, 3, H, W) batch by `ratio`, then
    grey (0.447) padding on the right / bottom up to the next multiple of `gs` of the scaled size (same_shape: up to the old size).
    Image preprocessing on the caller's device with stock torch ops — not part of the HIP path."""
    if ratio == 1.0:
        return img
    old_h, old_w = img.shape[2:]
    new_h, new_w = int(old_h * ratio), int(old_w * ratio)
    out = torch.nn.functional.interpolate(img, size=(new_h, new_w), mode='bilinear', align_corners=False)
    if same_shape:
        tgt_h, tgt_w = old_h, old_w
    else:
        tgt_h, tgt_w = (math.ceil(v * ratio / gs) * gs for v in (old_h, old_w))
    return torch.nn.functional.pad(out, [0, tgt_w - new_w, 0, tgt_h - new_h], value=0.447)


def freeze_params(model, layers=()):
    """requires_grad = False for parameters whose name equals or starts with one of `layers` (+ '.')."""
    for k, v in model.named_parameters():
        if any(k == name or k.startswith(name + '.') for name in layers):
            LOGGER.info(f'freezing {k}')
            v.requires_grad = False
    return model


class FrozenBatchNorm2d(nn.Module):
    """BatchNorm2d with fixed statistics and affine parameters (all four are buffers) — what the reference takes from
    torchvision.ops.FrozenBatchNorm2d (utils_torch.py:192).  It only holds state here: inside a launch plan it becomes a constant
    per-channel scale / shift in training as well as in eval (hd_yolo_amd/plan.py)."""

    def __init__(self, num_features, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.register_buffer('weight', torch.ones(num_features))
        self.register_buffer('bias', torch.zeros(num_features))
        self.register_buffer('running_mean', torch.zeros(num_features))
        self.register_buffer('running_var', torch.ones(num_features))

    def _load_from_state_dict(self, state_dict, prefix, *args):
        state_dict.pop(prefix + 'num_batches_tracked', None)
        super()._load_from_state_dict(state_dict, prefix, *args)

    def forward(self, x):
        raise RuntimeError('FrozenBatchNorm2d runs inside the HIP plan of its Conv block')


def freeze_bn(model, layers=()):
    """Replace every nn.BatchNorm2d under the named layers by a FrozenBatchNorm2d with the same statistics (reference :180-203);
    the launch plans are rebuilt on the next forward."""
    if not layers:
        return model
    swap = {}
    for k, m in model.named_modules():
        if isinstance(m, nn.BatchNorm2d) and any(k == name or k.startswith(name + '.') for name in layers):
            f = FrozenBatchNorm2d(m.num_features, m.eps).to(m.weight.device)
            f.load_state_dict(m.state_dict())
            swap[k] = f
    for k, f in swap.items():
        parent = model
        *path, leaf = k.split('.')
        for a in path:
            parent = getattr(parent, a)
        LOGGER.info(f'Replace layer: {k} with FrozenBatchNorm2d')
        setattr(parent, leaf, f)
    return model


class EarlyStopping:
    """Stop when fitness has not improved for `patience` epochs (reference :140-160)."""

    def __init__(self, patience=30):
        self.best_fitness, self.best_epoch = 0.0, 0
        self.patience = patience or float('inf')
        self.possible_stop = False

    def __call__(self, epoch, fitness):
        if fitness >= self.best_fitness:
            self.best_epoch, self.best_fitness = epoch, fitness
        delta = epoch - self.best_epoch
        self.possible_stop = delta >= (self.patience - 1)
        stop = delta >= self.patience
        if stop:
            LOGGER.info(f'Stopping training early as no improvement observed in last {self.patience} epochs.')
        return stop


def one_hot_labels(x, num_classes=None):
    """Labels 1..num_classes -> (N, num_classes+1) one-hot; anything else lands in column 0."""
    num_classes = num_classes or int(x.max())
    x = torch.where((x > 0) & (x <= num_classes), x, torch.zeros_like(x))
    return torch.nn.functional.one_hot(x, num_classes=num_classes + 1)
