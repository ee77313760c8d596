"""CPU restatement (torch fp32) of hnet's semantic-segmentation header — TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(),
bench.py's cpu_baseline may import it; the product path never does).

  connector(params, feats)            hnet/segmentation/utils_seg.py:18-36,56-59  per level: [conv3x3 (no bias) -> GroupNorm(32) -> ReLU
                                      (-> Upsample(2, bilinear, align_corners=True))] ladders, outputs summed
  panoptic(params, feats, cfg, masks) hnet/segmentation/panoptic_seg.py:13-19,28-41  Upsample(scale_factor) -> Conv1x1 -> Softmax2d
                                      -> F.interpolate(size of the masks, bilinear, align_corners=True) -> 1 + criterion(res, masks)
  soft_dice_criterion                 `SoftDiceLoss` is referenced by the reference (panoptic_seg.py:22) but defined nowhere in its
                                      repository: PARITY UNPINNED for this one function.  Restated from the repository's own dice,
                                      mask_iou(factor=0) (metayolo/models/utils_general.py:268-280): dice = 2*sum(t*p) / sum(t+p) over
                                      the pixels of one (image, class); criterion = -sum_c w_c mean_n dice / sum_c w_c.
Pinned by tests/golden/seg.npz, which tests/golden/make_golden.py writes by running the reference's own PanopticFeatureConnector and
PanopticSeg classes (with this file's criterion injected for the missing name)."""
import torch
import torch.nn.functional as F


def ladder_layout(nlevels):
    """[(has_upsample per stage)] per level, as PanopticFeatureConnector.__init__ builds them: level idx has max(idx, 1) stages"""
    out = []
    for idx in range(nlevels):
        out.append([False] if idx == 0 else [True] * idx)
    return out


def connector(params, feats, prefix=''):
    """params: state_dict of a PanopticFeatureConnector; feats: ordered {name: (N, C, h, w)} finest level first."""
    res = []
    for idx, (name, x) in enumerate(feats.items()):
        i = 0
        for up in ladder_layout(len(feats))[idx]:
            base = f'{prefix}layers.{name}.{i}'
            x = F.conv2d(x, params[f'{base}.weight'], None, 1, 1)
            x = F.group_norm(x, 32, params[f'{prefix}layers.{name}.{i + 1}.weight'], params[f'{prefix}layers.{name}.{i + 1}.bias'], 1e-5)
            x = F.relu(x)
            i += 3
            if up:
                x = F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=True)
                i += 1
        res.append(x)
    return sum(res)


def soft_dice_criterion(probs, masks, class_weight=None):
    prod, plus = (masks * probs).sum((2, 3)), (masks + probs).sum((2, 3))
    dice = (2 * prod / plus).mean(0)
    w = torch.ones_like(dice) if class_weight is None else torch.as_tensor(class_weight, dtype=dice.dtype)
    return -(dice * w).sum() / w.sum()


def panoptic(params, feats, scale_factor, masks=None, class_weight=None):
    """params: state_dict of a PanopticSeg ('connector.*', 'layers.<i>.weight|bias').  Returns (probabilities, loss or None)."""
    x = connector(params, feats, prefix='connector.')
    conv_i = 0
    if scale_factor is not None and scale_factor != 1:
        x = F.interpolate(x, scale_factor=scale_factor, mode='bilinear', align_corners=True)
        conv_i = 1
    x = F.conv2d(x, params[f'layers.{conv_i}.weight'], params[f'layers.{conv_i}.bias'])
    res = torch.softmax(x, 1)
    loss = None
    if masks is not None:
        res = F.interpolate(res, size=masks.shape[-2:], mode='bilinear', align_corners=True)
        loss = 1 + soft_dice_criterion(res, masks, class_weight)
    return res, loss
