"""CPU restatement of the mask branch's third-party pieces (TEST INFRASTRUCTURE ONLY — see oracle/__init__.py).

The reference takes these from packages that are not under /root/reference and that it pins no version of:
  torchvision.ops.roi_align                                   (metayolo/models/yolo_head.py:243, :294)
  torchvision.models.detection.mask_rcnn.MaskRCNNHeads / MaskRCNNPredictor   (yolo_head.py:11, :125-128)
  torch_scatter.scatter_max                                   (yolo_head.py:9, :257)
so, as for NMS, parity at this boundary is UNPINNED: the functions below restate the published algorithms (torchvision's
roi_align CPU kernel: sampling_ratio^2 bilinear samples per bin, `aligned` half-pixel shift, samples outside [-1, size] are zero,
coordinates clamped at 0, last row/column interpolates with itself; Mask R-CNN head = 4 x (conv3x3 + ReLU) then
ConvTranspose2d(2, 2) + ReLU + conv1x1; scatter_max = per-group maximum and its first arg-max, empty groups -> (0, len(src))).
They are plugged into the reference's own Detect (tests/golden/make_golden.py) to produce the mask goldens, and the GPU tests hold
the HIP kernels to them."""
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F


def roi_align(input, boxes, output_size, spatial_scale=1.0, sampling_ratio=-1, aligned=False):
    """input (B, C, H, W); boxes (R, 5) [image, x1, y1, x2, y2] or a list of (n_i, 4) per image -> (R, C, P, P)."""
    if isinstance(boxes, (list, tuple)):
        boxes = torch.cat([F.pad(b, [1, 0], value=float(i)) for i, b in enumerate(boxes)])
    P = output_size if isinstance(output_size, int) else output_size[0]
    B, C, H, W = input.shape
    R = boxes.shape[0]
    out = input.new_zeros((R, C, P, P))
    if R == 0:
        return out
    assert sampling_ratio > 0, 'adaptive sampling is not used by the reference'
    S = sampling_ratio
    off = 0.5 if aligned else 0.0
    bidx = boxes[:, 0].long()
    x0 = boxes[:, 1] * spatial_scale - off
    y0 = boxes[:, 2] * spatial_scale - off
    rw = boxes[:, 3] * spatial_scale - off - x0
    rh = boxes[:, 4] * spatial_scale - off - y0
    if not aligned:
        rw, rh = rw.clamp(min=1.0), rh.clamp(min=1.0)
    bw, bh = rw / P, rh / P
    grid = (torch.arange(P * S, dtype=input.dtype) // S).to(input.dtype), (torch.arange(P * S) % S).to(input.dtype)
    # sample coordinates (R, P*S)
    ys = y0[:, None] + grid[0][None] * bh[:, None] + (grid[1][None] + 0.5) * bh[:, None] / S
    xs = x0[:, None] + grid[0][None] * bw[:, None] + (grid[1][None] + 0.5) * bw[:, None] / S

    def prep(v, size):
        ok = ~((v < -1.0) | (v > size))
        v = v.clamp(min=0.0)
        lo = v.floor().long()
        hi_edge = lo >= size - 1
        lo = torch.where(hi_edge, torch.full_like(lo, size - 1), lo)
        hi = torch.where(hi_edge, lo, lo + 1)
        v = torch.where(hi_edge, lo.to(v.dtype), v)
        frac = v - lo.to(v.dtype)
        return ok, lo, hi, frac

    oky, ylo, yhi, ly = prep(ys, H)
    okx, xlo, xhi, lx = prep(xs, W)
    for r in range(R):
        f = input[bidx[r]]                                    # (C, H, W)
        wy0, wy1 = (1 - ly[r]) * oky[r], ly[r] * oky[r]       # (PS,)
        wx0, wx1 = (1 - lx[r]) * okx[r], lx[r] * okx[r]
        rows0, rows1 = f[:, ylo[r]], f[:, yhi[r]]             # (C, PS, W)
        v = (wy0[None, :, None] * wx0[None, None, :]) * rows0[:, :, xlo[r]] + (wy0[None, :, None] * wx1[None, None, :]) * rows0[:, :, xhi[r]] \
            + (wy1[None, :, None] * wx0[None, None, :]) * rows1[:, :, xlo[r]] + (wy1[None, :, None] * wx1[None, None, :]) * rows1[:, :, xhi[r]]
        out[r] = v.view(C, P, S, P, S).mean((2, 4))
    return out


class MaskRCNNHeads(nn.Sequential):
    """torchvision.models.detection.mask_rcnn.MaskRCNNHeads (classic naming: mask_fcn{i}, relu{i})."""

    def __init__(self, in_channels, layers, dilation):
        d = OrderedDict()
        c = in_channels
        for i, k in enumerate(layers, 1):
            d[f'mask_fcn{i}'] = nn.Conv2d(c, k, kernel_size=3, stride=1, padding=dilation, dilation=dilation)
            d[f'relu{i}'] = nn.ReLU(inplace=True)
            c = k
        super().__init__(d)
        for name, p in self.named_parameters():
            if 'weight' in name:
                nn.init.kaiming_normal_(p, mode='fan_out', nonlinearity='relu')


class MaskRCNNPredictor(nn.Sequential):
    def __init__(self, in_channels, dim_reduced, num_classes):
        super().__init__(OrderedDict([
            ('conv5_mask', nn.ConvTranspose2d(in_channels, dim_reduced, 2, 2, 0)),
            ('relu', nn.ReLU(inplace=True)),
            ('mask_fcn_logits', nn.Conv2d(dim_reduced, num_classes, 1, 1, 0)),
        ]))
        for name, p in self.named_parameters():
            if 'weight' in name:
                nn.init.kaiming_normal_(p, mode='fan_out', nonlinearity='relu')


def scatter_max(src, index, dim_size=None):
    """(per-group max, index of its first occurrence); empty groups: (0, len(src)) as torch_scatter does."""
    n = int(index.max()) + 1 if dim_size is None and index.numel() else (dim_size or 0)
    out = src.new_zeros(n)
    arg = torch.full((n,), src.numel(), dtype=torch.long)
    for g in range(n):
        sel = (index == g).nonzero().flatten()
        if sel.numel():
            v = src[sel]
            j = int(v.argmax())
            out[g], arg[g] = v[j], sel[j]
    return out, arg
