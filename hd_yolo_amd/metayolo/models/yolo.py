"""`Model` with the reference's constructor, attributes, forward signature and return structure
(reference: metayolo/models/yolo.py:26-107).

forward(x, targets=None, visualize=False, compute_masks=False) -> (losses, outputs)
    x        (B, 3, H, W) float tiles in 0..1 on the MI355X
    losses   {task: {'det_loss', 'mask_loss', 'loss_items': {'box','obj','cls','mask'}}}   (when targets are given)
    outputs  list over images of {task: {'boxes' (n,4) xyxy px, 'scores' (n,), 'labels' (n,) 1..nc or -100}}   (eval)

Backbone, neck and every header's 1x1 detection convs execute as ONE static HIP plan (hd_yolo_amd/plan.py); the plan's
fp32 logits enter autograd through a single custom Function, so `loss.backward()` replays the plan's backward launch list
and leaves parameter gradients as views of one flat buffer.
Arithmetic type: bf16 (fp32 accumulate) under torch autocast or after .half()/.bfloat16(); exact fp32 otherwise.
"""
from copy import deepcopy
from typing import Any, Dict, List, Union

import torch
import torch.nn as nn

from .. import LOGGER, check_version, load_cfg  # noqa: F401
from ... import engine as _engine
from .utils_general import make_divisible, nms  # noqa: F401
from .utils_torch import freeze_bn, freeze_params, fuse_conv_and_bn, initialize_weights, model_info, scale_img  # noqa: F401
from .yolov5 import *  # noqa: F401,F403
from .yolov5 import Conv, Detect, build_network


class Model(nn.Module):
    def __init__(self, cfg='yolov5s.yaml', hyp='./hyp.scratch.yaml', ch=3, anchors=None, is_scripting=False):
        super().__init__()
        self.cfg = deepcopy(load_cfg(cfg))
        self.hyp = deepcopy(load_cfg(hyp))
        self.inplace = self.cfg.get('inplace', True)
        self.cfg['ch'] = self.cfg.get('ch', ch)
        self.amp = self.cfg.get('amplification', None)
        if anchors:
            LOGGER.info(f'Overriding model.cfg anchors with anchors={anchors}')
            self.cfg['anchors'] = round(anchors)
        self.backbone, self.neck, self.headers = build_network(self.cfg, self.hyp, is_scripting=is_scripting)
        initialize_weights(self)
        self.info()
        LOGGER.info('')

    # ------------------------------------------------------------------ engine
    def _eng(self):
        eng = self.__dict__.get('_hdy_engine')
        if eng is None:
            heads = list(self.headers.values())
            eng = _engine.Engine(self.backbone, self.neck, heads[0] if len(heads) == 1 else heads)
            object.__setattr__(self, '_hdy_engine', eng)
        return eng

    def half(self):
        """The reference validates with model.half() (val_nuclei.py:116).  Parameters stay fp32 masters here; the call
        switches the plans to bf16 operands."""
        object.__setattr__(self, 'hdy_dtype', torch.bfloat16)
        return self

    bfloat16 = half

    def float(self):
        object.__setattr__(self, 'hdy_dtype', None)
        return super().float()

    def features(self, x):
        """{layer index: NCHW view} of the neck outputs for x (eval-mode statistics unless self.training)."""
        plan, _ = self._eng().forward(x, self.training, _engine.compute_dtype(self, x))
        return {k: plan.feature(k) for k in self.neck.save}

    # ------------------------------------------------------------------ forward
    def forward(self, x: torch.Tensor, targets=None, visualize=False, compute_masks=False):
        dtype = _engine.compute_dtype(self, x)
        if self.training and targets is not None and torch.is_grad_enabled() and len(self.headers) == 1:
            task_id, header = next(iter(self.headers.items()))
            if header.fused_loss_ok() and all(task_id in t['anns'] and len(t['anns'][task_id]) == 1 for t in targets):
                # forward launch list, then target assignment + loss + logits gradient in one fused launch sequence
                gts = [t['anns'][task_id][0] for t in targets]
                _, losses = header.fused_losses(self._eng(), x, dtype, gts, compute_masks=compute_masks)
                return {task_id: losses}, self.post_processing([{task_id: o} for o in []])
        plan, dets = self._eng().forward(x, self.training, dtype)
        losses, outputs = {}, {}
        first = 0
        for task_id, header in self.headers.items():
            # the plan lists the logits header after header (several headers: each runs its own matcher / DetLoss / outputs on its
            # levels; the shared backbone and neck receive the sum of their gradients — reference: yolo.py:62-81)
            nl = len(header.m)
            task_dets, task_gts = dets[first:first + nl], None
            first += nl
            all_dets = task_dets
            if targets is not None:
                task_gts, keep = [], []
                for idx, t in enumerate(targets):
                    if task_id in t['anns']:
                        task_gts.extend(t['anns'][task_id])
                        keep.extend([idx] * len(t['anns'][task_id]))
                if keep != list(range(x.shape[0])):
                    sel = torch.tensor(keep, device=x.device, dtype=torch.long)
                    task_dets = [d.index_select(0, sel) for d in all_dets]
            mask_ctx = (self._eng(), plan, dtype) if getattr(header, 'nc_masks', 0) > 0 and len(self.headers) == 1 else None
            losses[task_id], outputs[task_id] = header.forward_dets(task_dets, task_gts, compute_masks=compute_masks, mask_ctx=mask_ctx)
        outputs = [dict(zip(outputs.keys(), per_image)) for per_image in zip(*outputs.values())]
        return losses, self.post_processing(outputs)

    def post_processing(self, outputs: Any):
        return outputs

    def fuse(self):
        """Fold every BatchNorm into its conv (deploy); the plans are rebuilt for the folded graph."""
        LOGGER.info('Fusing layers... ')
        for m in self.modules():
            if isinstance(m, Conv) and hasattr(m, 'bn'):
                m.conv = fuse_conv_and_bn(m.conv, m.bn)
                delattr(m, 'bn')
                m.forward = m.forward_fuse
        self.__dict__.pop('_hdy_engine', None)
        self.info()
        return self

    def info(self, verbose=False, img_size=640):
        model_info(self, verbose, img_size)

    def freeze(self, layers=[]):
        freeze_params(self, layers)
        freeze_bn(self, layers)
        return self


class Deploy(nn.Module):
    """Inference wrapper with the reference's call signature (reference: metayolo/models/yolo.py:110-142):
    forward(x, compute_masks=True) -> (None, [ {task: {'boxes','scores','labels'}} per image ]).

    The reference freezes backbone / neck / headers into TorchScript here; this build has nothing to script — the eval
    launch list of the wrapped Model IS the deployed artefact (static HIP plan per input shape, BN folded into the conv
    epilogue whether or not `fuse` is asked for).  `fuse=True` additionally folds the parameters themselves, as the reference
    does, on a copy."""

    def __init__(self, model, fuse=False):
        super().__init__()
        if fuse:
            model = deepcopy(model).fuse()          # launch plans are per model object and are not copied (engine.Engine)
        model.eval()
        object.__setattr__(self, '_model', model)                # not a registered child: the module tree is the reference's
        self.backbone, self.neck, self.headers = model.backbone, model.neck, model.headers

    def half(self):
        self._model.half()
        return self

    def float(self):
        self._model.float()
        return self

    @torch.no_grad()
    def forward(self, x: Union[List[torch.Tensor], torch.Tensor], compute_masks: bool = True):
        if isinstance(x, (list, tuple)):
            x = torch.stack(list(x))
        _, outputs = self._model(x, compute_masks=compute_masks)
        return None, self.post_processing(outputs)

    def post_processing(self, x: Any):
        return x


class Ensemble(nn.ModuleList):
    """Several models on the same tiles; per image and task their detections are pooled, thresholded and passed through one
    class-agnostic NMS (reference: metayolo/models/yolo.py:145-204).  The NMS is the MI355X kernel (hdy_nms_boxes)."""

    def __init__(self, models, nms_params: Dict[str, float] = {}):
        super().__init__(models)
        self.nms_params = self.get_nms_params(nms_params)

    def get_nms_params(self, args={}):
        defaults = {'conf_thres': 0.15, 'iou_thres': 0.45, 'max_det': 300}
        return {k: float(args.get(k, v)) for k, v in defaults.items()}

    @torch.no_grad()
    def forward(self, x: Union[List[torch.Tensor], torch.Tensor], compute_masks: bool = True):
        if isinstance(x, (list, tuple)):
            x = torch.stack(list(x))
        per_model = [m(x, compute_masks=compute_masks)[1] for m in self]
        return None, [self.merge([o[i] for o in per_model]) for i in range(x.shape[0])]

    def merge(self, x: List[Dict[str, Any]]):
        res = {}
        for task_id in sorted(set().union(*x)):
            parts = [r[task_id] for r in x if task_id in r]
            boxes = torch.cat([q['boxes'] for q in parts])
            scores = torch.cat([q['scores'] for q in parts])
            labels = torch.cat([q['labels'] for q in parts])
            masks = None
            with_masks = [q['masks'] for q in parts if 'masks' in q]
            if with_masks:
                # masks ride along with their boxes (reference: yolo.py:172-186).  A model that returned none contributes zero masks for
                # ITS boxes; the reference, marked "not finished" there, appends a single zero mask instead and then fails on the
                # length mismatch whenever that model had a different number of boxes.
                m0 = with_masks[0]
                masks = torch.cat([q['masks'] if 'masks' in q else m0.new_zeros((len(q['boxes']),) + tuple(m0.shape[1:])) for q in parts])
            sel = scores > self.nms_params['conf_thres']
            boxes, scores, labels = boxes[sel], scores[sel], labels[sel]
            if masks is not None:
                masks = masks[sel]
            if len(boxes):
                keep = nms(boxes, scores, self.nms_params['iou_thres'])[:int(self.nms_params['max_det'])]
                boxes, scores, labels = boxes[keep], scores[keep], labels[keep]
                if masks is not None:
                    masks = masks[keep]
            res[task_id] = {'boxes': boxes, 'scores': scores, 'labels': labels}
            if masks is not None:
                res[task_id]['masks'] = masks
        return res
