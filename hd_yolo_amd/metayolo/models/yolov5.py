"""YAML -> module tree, with the reference's container classes and config schema
(reference: metayolo/models/yolov5.py:47-59 CSPDarkNet, :62-77 FPN, :80-161 build_network).

The containers keep their Sequential structure (so state_dict keys are `backbone.{i}...`, `neck.{i}...`) and the routing
attributes build_network attaches (`.i .f .type .np .tag`).  Their own forward runs a HIP plan over the container; inside
`Model` a single plan spans backbone + neck + detection convs instead.
"""
from typing import Dict, Iterable, List, Optional

import torch
import torch.nn as nn

from .. import LOGGER, check_version, load_cfg  # noqa: F401
from ... import engine as _engine
from .layers import *  # noqa: F401,F403
from .layers import C3, SPPF, Bottleneck, Concat, Conv
from .utils_general import make_divisible
from .utils_torch import fuse_conv_and_bn, initialize_weights, model_info, scale_img  # noqa: F401
from .yolo_head import Detect

_MODULES = {'Conv': Conv, 'Bottleneck': Bottleneck, 'C3': C3, 'SPPF': SPPF, 'Concat': Concat, 'Detect': Detect,
            'nn.Upsample': nn.Upsample, 'nn.BatchNorm2d': nn.BatchNorm2d}
_WIDTH_SCALED = (Conv, Bottleneck, SPPF, C3)
_REPEAT_INSIDE = (C3,)


class CSPDarkNet(nn.Sequential):
    def __init__(self, modules: Optional[Iterable[nn.Module]] = None, return_layers: Optional[List] = None) -> None:
        super().__init__(*modules)
        self.save = return_layers or [len(self) - 1]

    def forward(self, x: torch.Tensor) -> Dict[int, torch.Tensor]:
        eng = self.__dict__.get('_hdy_engine')
        if eng is None:
            eng = _engine.Engine(self)
            object.__setattr__(self, '_hdy_engine', eng)
        plan = eng.plan_for(x, False, _engine.compute_dtype(self, x))
        plan.run_forward(x)
        return {k: plan.feature(k) for k in self.save}


class FPN(nn.Sequential):
    def __init__(self, modules: Optional[Iterable[nn.Module]] = None, return_layers: Optional[List] = None) -> None:
        super().__init__(*modules)
        self.save = return_layers or [len(self) - 1]

    def forward(self, x: Dict[int, torch.Tensor]) -> Dict[int, torch.Tensor]:
        """Neck on bare backbone features {layer index: NCHW} (eval statistics; training runs inside the Model plan).  Like the
        reference (yolov5.py:68-77) the input dict is updated with the saved layers and `-1`."""
        if self.training and torch.is_grad_enabled():
            raise RuntimeError('FPN.forward on bare features is forward-only: train through Model (one launch list, one backward)')
        eng = self.__dict__.get('_hdy_engine')
        if eng is None:
            eng = _engine.Engine(None, self, None)
            object.__setattr__(self, '_hdy_engine', eng)
        feats = {k: v for k, v in x.items() if isinstance(k, int) and k >= 0}
        first = next(iter(feats.values()))
        plan = eng.plan_for_features(feats, _engine.compute_dtype(self, first))
        plan.run_forward_features(feats)
        for k in self.save:
            x[k] = plan.feature(k)
        x[-1] = plan.feature(self[-1].i)
        return {k: x[k] for k in self.save}


def _resolve(name):
    if not isinstance(name, str):
        return name
    if name not in _MODULES:
        raise NotImplementedError(f"module '{name}' is not part of the metayolo detection hot path on MI355X "
                                  f"(supported: {sorted(_MODULES)})")
    return _MODULES[name]


def build_network(cfg, hyp, is_scripting=False):
    """rows of cfg['backbone'] + cfg['fpn'] + cfg['headers']: [from, number, module, args, (tag), (header args)]."""
    LOGGER.info(f"\n{'':>3}{'from':>18}{'n':>3}{'params':>10}  {'module':<40}{'arguments':<30}")
    gd, gw = cfg['depth_multiple'], cfg['width_multiple']
    ch = [cfg['ch']]
    layers, save, c2 = [], [], ch[-1]
    for i, row in enumerate(cfg['backbone'] + cfg['fpn'] + cfg['headers']):
        f, n, m, args = row[0], row[1], _resolve(row[2]), list(row[3])
        tag = row[4] if len(row) > 4 else None
        args = [cfg[a] if isinstance(a, str) and a in cfg else a for a in args]
        n = n_ = max(round(n * gd), 1) if n > 1 else n
        if m is Detect:
            tag = tag or 'det'
            args = [[ch[x] for x in f]] + args
            if isinstance(args[1], int):
                args[1] = [list(range(args[1] * 2))] * len(f)
            h = hyp[tag]
            loss_keys = ('box', 'cls', 'cls_pw', 'cls_cw', 'obj', 'obj_pw', 'mask', 'iou_t', 'anchor_t', 'fl_gamma', 'label_smoothing')
            loss_hyp = {k: h[k] for k in loss_keys if k in h}
            nms_params = {k: h[k] for k in ('conf_thres', 'iou_thres', 'max_det') if k in h}
            if isinstance(args[-1], int):
                args[-1] = {c: args[-1] for c in range(args[-2] + 1)}
            m_ = m(*args, multi_label=bool(h['multi_label']), nms_params=nms_params, loss_hyp=loss_hyp, is_scripting=is_scripting)
        else:
            if m in _WIDTH_SCALED:
                c1, c2 = ch[f], make_divisible(args[0] * gw, 8)
                args = [c1, c2, *args[1:]]
                if m in _REPEAT_INSIDE:
                    args.insert(2, n)
                    n = 1
            elif m is nn.BatchNorm2d:
                args = [ch[f]]
            elif m is Concat:
                c2 = sum(ch[x] for x in f)
            else:
                c2 = ch[f]
            m_ = nn.Sequential(*(m(*args) for _ in range(n))) if n > 1 else m(*args)
        t = str(m)[8:-2].replace('__main__.', '')
        np_ = sum(x.numel() for x in m_.parameters())
        m_.i, m_.f, m_.type, m_.np, m_.tag = i, f, t, np_, tag
        LOGGER.info(f'{i:>3}{str(f):>18}{n_:>3}{np_:10.0f}  {t:<40}{str(args):<30}')
        save.extend(x % i for x in ([f] if isinstance(f, int) else f) if x != -1)
        layers.append(m_)
        if i == 0:
            ch = []
        ch.append(c2)
    save = sorted(save)
    n1, n2 = len(cfg['backbone']), len(cfg['fpn'])
    backbone = CSPDarkNet(layers[:n1], [s for s in save if s < n1])
    fpn = FPN(layers[n1:n1 + n2], save)
    headers = nn.ModuleDict({m.tag: m for m in layers[n1 + n2:]})
    return backbone, fpn, headers
