"""Parameter containers of the Mask R-CNN mask head with torchvision's (classic) module and parameter names, so that reference
checkpoints load key-for-key: `seg_h.maskrcnn_heads.mask_fcn{1..4}.{weight,bias}`, `seg_h.maskrcnn_preds.conv5_mask.*`,
`seg_h.maskrcnn_preds.mask_fcn_logits.*` (reference: metayolo/models/yolo_head.py:11, :125-128 — torchvision is not installed here).
The arithmetic runs in hd_yolo_amd/maskhead.py on the HIP kernels; calling these modules directly raises."""
from collections import OrderedDict

import torch.nn as nn


class _HipOnly(nn.Sequential):
    def forward(self, x):
        raise RuntimeError(f'{type(self).__name__} holds parameters only: the mask head runs through Detect on the MI355X '
                           '(hd_yolo_amd/maskhead.py)')


class MaskRCNNHeads(_HipOnly):
    def __init__(self, in_channels, layers, dilation):
        if dilation != 1:
            raise NotImplementedError('dilated mask heads are not on the hot path')
        d = OrderedDict()
        c = in_channels
        for i, k in enumerate(layers, 1):
            d[f'mask_fcn{i}'] = nn.Conv2d(c, k, kernel_size=3, stride=1, padding=1)
            d[f'relu{i}'] = nn.ReLU(inplace=True)
            c = k
        super().__init__(d)
        for name, p in self.named_parameters():
            if 'weight' in name:
                nn.init.kaiming_normal_(p, mode='fan_out', nonlinearity='relu')


class MaskRCNNPredictor(_HipOnly):
    def __init__(self, in_channels, dim_reduced, num_classes):
        super().__init__(OrderedDict([
            ('conv5_mask', nn.ConvTranspose2d(in_channels, dim_reduced, 2, 2, 0)),
            ('relu', nn.ReLU(inplace=True)),
            ('mask_fcn_logits', nn.Conv2d(dim_reduced, num_classes, 1, 1, 0)),
        ]))
        for name, p in self.named_parameters():
            if 'weight' in name:
                nn.init.kaiming_normal_(p, mode='fan_out', nonlinearity='relu')
