"""HNet: one backbone + feature pyramid shared by several task headers, mixed loss, one backward (reference: hnet/hnet.py:104-265).

    configs = {'backbone': {'type': 'yolov5', 'cfg': <metayolo model cfg: backbone / fpn / headers>, 'hyp': <hyp dict>},
               'headers': {'seg': {'type': 'PanopticSeg', 'configs': {'num_classes': 3, 'feature_maps': None, 'in_channels': None,
                                                                      'scale_factor': 8, 'resize_mode': 'bilinear', 'class_weight': None,
                                                                      'roi_size': None}}}}
    forward(images, targets=None) -> (losses, outputs)        (reference :189-255)
        images   (N, 3, H, W) float tiles in 0..1 on the GPU (the reference's GeneralizedTransform list -> batch step is data plumbing)
        targets  per image {'anns': {task_id: [annotation, ...]}}: detection tasks as metayolo (boxes / labels), PanopticSeg tasks
                 {'roi': (4,), 'masks': (num_classes, H, W)}
        losses   {f'{task_id}_{name}': tensor}, e.g. 'det_det_loss', 'seg_soft_iou_loss'   (reference :240)
        outputs  {task_id: per-image results}

The reference builds its backbone with timm / Swin and its detection header with its own Mask R-CNN; neither package exists here,
and the file pins sub-modules to cuda:0 / cuda:2 (hnet/hnet.py:178-180).  Here the backbone, pyramid and detection header(s) are a
metayolo `Model` ('yolov5' type) — ONE static HIP plan for conv stack + detection convs, with the pyramid levels the segmentation
headers read exposed as autograd outputs of that plan (engine taps) — and the PanopticSeg headers run on csrc/seg.hip."""
from collections import OrderedDict

import torch

from .. import _lib
from .. import engine as _engine
from ..metayolo.models.yolo import Model
from .segmentation import PanopticSeg


class HNet(torch.nn.Module):
    def __init__(self, configs):
        super().__init__()
        self.configs = configs
        bb = configs['backbone']
        if bb.get('type', 'yolov5') != 'yolov5':
            raise _lib.HdyError(f"HNet backbone type {bb['type']!r}: only the metayolo conv backbone ('yolov5') exists on this path")
        self.detector = Model(bb['cfg'], bb['hyp'])
        self.backbone, self.fpn = self.detector.backbone, self.detector.neck
        det_head = next(iter(self.detector.headers.values()))
        levels = list(det_head.f)                                        # pyramid layers, finest first (e.g. [17, 20, 23])
        chans = [m.in_channels for m in det_head.m]
        self.headers = torch.nn.ModuleDict()
        self.roi_sizes, self.feature_maps = {}, {}
        for task_id, cfg in configs.get('headers', {}).items():
            c = cfg['configs']
            if cfg['type'] != 'PanopticSeg':
                raise NotImplementedError(f"header type {cfg['type']}: detection headers come with the backbone cfg; only PanopticSeg is added here")
            if c.get('feature_maps') is None:
                c['feature_maps'] = OrderedDict((str(k), str(k)) for k in levels)
            maps = [int(k) for k in c['feature_maps']]
            if c.get('in_channels') is None:
                c['in_channels'] = chans[levels.index(maps[0])]
            self.roi_sizes[task_id], self.feature_maps[task_id] = c.get('roi_size'), c['feature_maps']
            head = PanopticSeg(c)
            # the connector's first convolution of each level takes that level's width
            for name, k in zip(head.connector.layers, maps):
                conv = head.connector.layers[name][0]
                cin = chans[levels.index(k)]
                if conv.in_channels != cin:
                    head.connector.layers[name][0] = torch.nn.Conv2d(cin, conv.out_channels, 3, stride=1, padding=1, bias=False)
            self.headers[task_id] = head
        self.taps = sorted({int(k) for fm in self.feature_maps.values() for k in fm})

    def half(self):
        self.detector.half()
        return self

    bfloat16 = half

    def float(self):
        self.detector.float()
        return super().float()

    def _eng(self):
        eng = self.detector.__dict__.get('_hdy_engine')
        if eng is None or eng.taps != tuple(self.taps):
            heads = list(self.detector.headers.values())
            eng = _engine.Engine(self.backbone, self.fpn, heads[0] if len(heads) == 1 else heads, extra=list(self.headers.values()),
                                 taps=self.taps)
            object.__setattr__(self.detector, '_hdy_engine', eng)
        return eng

    def forward(self, images, targets=None):
        if self.training and targets is None:
            raise ValueError('In training mode, targets should be passed')
        if isinstance(images, (list, tuple)):
            images = torch.stack(list(images))
        eng = self._eng()                                                # the detector's forward below runs on this engine
        image_size = tuple(images.shape[-2:])
        det_targets = None
        if targets is not None:
            det_targets = [{**t, 'anns': {k: v for k, v in t['anns'].items() if k in self.detector.headers}} for t in targets]
        det_losses, det_outputs = self.detector(images, det_targets)
        plan = eng.last_plan
        feats = dict(zip(plan.tap_keys, eng.tap_outputs if eng.tap_outputs is not None else plan.tap_features()))
        dtype = _engine.compute_dtype(self.detector, images)
        losses, outputs = {}, {}
        for task_id, task_losses in (det_losses or {}).items():
            for k in ('det_loss', 'mask_loss'):
                if task_losses and k in task_losses:
                    losses[f'{task_id}_{k}'] = task_losses[k]
        for task_id in self.detector.headers:
            outputs[task_id] = [o[task_id] for o in det_outputs] if det_outputs else []
        for task_id, head in self.headers.items():
            task_feats = OrderedDict((k, feats[int(k)]) for k in self.feature_maps[task_id])
            task_targets = None
            if targets is not None:
                task_targets = [t['anns'].get(task_id, []) for t in targets]
            grad_of = eng.tap_grad_sink(plan) if (self.training and torch.is_grad_enabled()) else None
            task_out, task_losses = head(task_feats, image_size, self.roi_sizes[task_id], task_targets, grad_of=grad_of, dtype=dtype)
            losses.update({f'{task_id}_{k}': v for k, v in task_losses.items()})
            outputs[task_id] = task_out
        return losses, outputs
