"""PanopticFeatureConnector with the reference's constructor and parameter names (reference: hnet/segmentation/utils_seg.py:5-59):
per pyramid level a ladder of Conv3x3(no bias) -> GroupNorm(32) -> ReLU (-> 2x bilinear, align_corners=True) stages up to the finest
level's resolution, outputs summed.  forward() on GPU tensors runs hd_yolo_amd/segrun.py (HIP kernels); this module only owns the
parameters (`layers.<name>.<i>.weight|bias`)."""
import numbers

import torch

from ... import _lib, ops
from ...segrun import PackCache, PanopticRun

__all__ = ['PanopticFeatureConnector']


class PanopticFeatureConnector(torch.nn.Module):
    def __init__(self, in_channels, out_channel, feature_maps, mode='bilinear'):
        super().__init__()
        if isinstance(in_channels, numbers.Number):
            in_channels = [in_channels] * len(feature_maps)
        if mode != 'bilinear':
            raise _lib.HdyError('PanopticFeatureConnector: only bilinear upsampling has a HIP kernel')
        self.in_channels, self.out_channel, self.feature_maps = in_channels, out_channel, feature_maps
        self.layers = torch.nn.ModuleDict()
        for idx, (in_c, name) in enumerate(zip(in_channels, feature_maps)):
            blocks = [torch.nn.Conv2d(in_c, out_channel, 3, stride=1, padding=1, bias=False),
                      torch.nn.GroupNorm(num_groups=32, num_channels=out_channel), torch.nn.ReLU(inplace=True)]
            if idx > 0:
                blocks.append(torch.nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True))
            for _ in range(idx - 1):
                blocks += [torch.nn.Conv2d(out_channel, out_channel, 3, stride=1, padding=1, bias=False),
                           torch.nn.GroupNorm(num_groups=32, num_channels=out_channel), torch.nn.ReLU(inplace=True),
                           torch.nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)]
            self.layers[str(name)] = torch.nn.Sequential(*blocks)

    def forward(self, features):
        """{name: NCHW tensor} -> {'0': summed NCHW map}; inference-style call (no autograd): training goes through PanopticSeg."""
        feats = [features[k] for k in self.layers.keys()]
        ops.require_gpu(feats[0])
        dt = torch.bfloat16 if feats[0].dtype in (torch.bfloat16, torch.float16) else torch.float32
        cache = self.__dict__.setdefault('_hdy_pack', PackCache())
        run = PanopticRun(self, _Identity1x1(self.out_channel, feats[0].device), dt, cache)
        run.forward([f.to(dt).permute(0, 2, 3, 1).contiguous() for f in feats])
        return {'0': run.total.permute(0, 3, 1, 2)}


class _Identity1x1:
    """stand-in class head for the connector-only call (its logits are discarded)"""

    def __init__(self, c, device):
        self.out_channels = 8
        self.weight = torch.zeros((8, c, 1, 1), device=device)
        self.bias = torch.zeros(8, device=device)
