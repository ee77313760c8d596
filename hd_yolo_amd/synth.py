"""Deterministic synthetic configs, weights, tiles and targets.

Shared by the golden-vector generator (tests/golden/make_golden.py), the parity
tests and bench.py, so that every side of a comparison sees the same numbers.
Nothing here depends on module construction order: values are a function of
(key name, shape, seed) only.

Reference anchors: model yaml schema `metayolo/models/yolov5.py:80-161`
(SURVEY.md Appendix A), hyp sub-dict `metayolo/models/yolov5.py:105-110`,
target schema `metayolo/datasets.py:462-519`.
"""
import zlib
from copy import deepcopy

import torch

# depth_multiple, width_multiple of the stock hub files
# (metayolo/hub/yolov5{n,s,m,l}.yaml:5-6)
VARIANTS = {'n': (0.33, 0.25), 's': (0.33, 0.50), 'm': (0.67, 0.75), 'l': (1.0, 1.0)}

ANCHORS_P5 = [[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]]
# the 4-level anchors of the reference's own metayolo-schema files (metayolo/hub/yolov5m6-multihead.yaml:7-11)
ANCHORS_P6 = [[19, 27, 44, 40, 38, 94], [96, 68, 86, 152, 180, 137], [140, 301, 303, 264, 238, 542], [436, 615, 739, 380, 925, 792]]


def make_cfg(variant='s', nc=8):
    """metayolo-schema config (backbone + fpn + headers); SURVEY.md Appendix A.  'n' / 's' / 'm' / 'l' = 3-level P5 graph;
    a trailing '6' ('n6', 'm6', 'l6') = the 4-level P6 graph of the files the reference ships
    (metayolo/hub/yolov5{m6,l6}-multihead.yaml, yolov5l6-mask.yaml)."""
    if variant.endswith('6'):
        return make_cfg_p6(variant[:-1], nc)
    gd, gw = VARIANTS[variant]
    cfg = {
        'depth_multiple': gd, 'width_multiple': gw,
        'anchors': deepcopy(ANCHORS_P5),
        'backbone': [
            [-1, 1, 'Conv', [64, 6, 2, 2]],
            [-1, 1, 'Conv', [128, 3, 2]],
            [-1, 3, 'C3', [128]],
            [-1, 1, 'Conv', [256, 3, 2]],
            [-1, 6, 'C3', [256]],
            [-1, 1, 'Conv', [512, 3, 2]],
            [-1, 9, 'C3', [512]],
            [-1, 1, 'Conv', [1024, 3, 2]],
            [-1, 3, 'C3', [1024]],
            [-1, 1, 'SPPF', [1024, 5]],
        ],
        'fpn': [
            [9, 1, 'Conv', [512, 1, 1]],
            [-1, 1, 'nn.Upsample', [None, 2, 'nearest']],
            [[-1, 6], 1, 'Concat', [1]],
            [-1, 3, 'C3', [512, False]],
            [-1, 1, 'Conv', [256, 1, 1]],
            [-1, 1, 'nn.Upsample', [None, 2, 'nearest']],
            [[-1, 4], 1, 'Concat', [1]],
            [-1, 3, 'C3', [256, False], 'P3'],
            [-1, 1, 'Conv', [256, 3, 2]],
            [[-1, 14], 1, 'Concat', [1]],
            [-1, 3, 'C3', [512, False], 'P4'],
            [-1, 1, 'Conv', [512, 3, 2]],
            [[-1, 10], 1, 'Concat', [1]],
            [-1, 3, 'C3', [1024, False], 'P5'],
        ],
        'headers': [
            [[17, 20, 23], 1, 'Detect', ['anchors', [8.0, 16.0, 32.0], nc, -1], 'det'],
        ],
    }
    return cfg


def make_cfg_p6(variant='m', nc=8):
    """Layer list of metayolo/hub/yolov5m6-multihead.yaml:13-66 (P3-P6 outputs, strides 8-64)."""
    gd, gw = VARIANTS[variant]
    return {
        'depth_multiple': gd, 'width_multiple': gw,
        'anchors': deepcopy(ANCHORS_P6),
        'backbone': [
            [-1, 1, 'Conv', [64, 6, 2, 2]],
            [-1, 1, 'Conv', [128, 3, 2]],
            [-1, 3, 'C3', [128]],
            [-1, 1, 'Conv', [256, 3, 2]],
            [-1, 6, 'C3', [256]],
            [-1, 1, 'Conv', [512, 3, 2]],
            [-1, 9, 'C3', [512]],
            [-1, 1, 'Conv', [768, 3, 2]],
            [-1, 3, 'C3', [768]],
            [-1, 1, 'Conv', [1024, 3, 2]],
            [-1, 3, 'C3', [1024]],
            [-1, 1, 'SPPF', [1024, 5]],
        ],
        'fpn': [
            [11, 1, 'Conv', [768, 1, 1]],
            [-1, 1, 'nn.Upsample', [None, 2, 'nearest']],
            [[-1, 8], 1, 'Concat', [1]],
            [-1, 3, 'C3', [768, False]],
            [-1, 1, 'Conv', [512, 1, 1]],
            [-1, 1, 'nn.Upsample', [None, 2, 'nearest']],
            [[-1, 6], 1, 'Concat', [1]],
            [-1, 3, 'C3', [512, False]],
            [-1, 1, 'Conv', [256, 1, 1]],
            [-1, 1, 'nn.Upsample', [None, 2, 'nearest']],
            [[-1, 4], 1, 'Concat', [1]],
            [-1, 3, 'C3', [256, False], 'P3'],
            [-1, 1, 'Conv', [256, 3, 2]],
            [[-1, 20], 1, 'Concat', [1]],
            [-1, 3, 'C3', [512, False], 'P4'],
            [-1, 1, 'Conv', [512, 3, 2]],
            [[-1, 16], 1, 'Concat', [1]],
            [-1, 3, 'C3', [768, False], 'P5'],
            [-1, 1, 'Conv', [768, 3, 2]],
            [[-1, 12], 1, 'Concat', [1]],
            [-1, 3, 'C3', [1024, False], 'P6'],
        ],
        'headers': [
            [[23, 26, 29, 32], 1, 'Detect', ['anchors', [8.0, 16.0, 32.0, 64.0], nc, -1], 'det'],
        ],
    }


def make_hyp(conf_thres=0.15, iou_thres=0.45, max_det=300, multi_label=False):
    return {
        'lr0': 0.01, 'lrf': 0.1, 'momentum': 0.937, 'weight_decay': 0.0005,
        'warmup_epochs': 3.0, 'warmup_momentum': 0.8, 'warmup_bias_lr': 0.1,
        'det': {'box': 0.05, 'cls': 0.5, 'cls_pw': 1.0, 'cls_cw': 1.0, 'obj': 1.0, 'obj_pw': 1.0,
                'mask': 1.0, 'iou_t': 0.2, 'anchor_t': 4.0, 'fl_gamma': 0.0, 'label_smoothing': 0.0,
                'conf_thres': conf_thres, 'iou_thres': iou_thres, 'max_det': max_det,
                'multi_label': multi_label},
    }


def _gen(key, seed):
    g = torch.Generator(device='cpu')
    g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def synth_state_dict(shapes, seed=0):
    """shapes: {state_dict key: (shape, dtype)} -> {key: tensor}, values a function of key.

    Conv weights are uniform with the fan-in bound of PyTorch's default init, BN affine
    and running statistics are perturbed away from (1, 0, 0, 1) so that BN folding and
    the running-stat update are actually exercised; integer buffers and the Detect
    geometry buffers (anchors, grids) are left out (the model keeps its own).
    """
    out = {}
    for key, (shape, dtype) in shapes.items():
        if not dtype.is_floating_point:
            continue
        if '.anchors.' in key or key.endswith('mask_indices') or 'det_loss' in key or 'seg_loss' in key:
            continue
        g = _gen(key, seed)
        u = torch.rand(shape, generator=g, dtype=torch.float32)
        if key.endswith('running_var'):
            v = 0.5 + u
        elif key.endswith('running_mean'):
            v = (u - 0.5) * 0.4
        elif '.bn.' in key and key.endswith('weight'):
            v = 0.5 + u
        elif '.bn.' in key and key.endswith('bias'):
            v = (u - 0.5) * 0.4
        elif '.seg_h.' in key and key.endswith('bias'):   # Mask R-CNN head: small positive biases keep the ReLU chain alive
            v = u * 0.2
        elif key.endswith('bias'):          # Detect 1x1 conv bias
            v = (u - 0.5) * 2.0 - 2.0
        else:                               # conv weight [K, C, R, S]
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            bound = (3.0 / fan_in) ** 0.5 * 1.4
            v = (u * 2 - 1) * bound
        out[key] = v.to(dtype)
    return out


def calibrate_det_logits(model, x, tag='det'):
    """Random-init deep variants (yolov5l: ~100 convolutions with eval-mode BatchNorm) blow the activations up by orders of magnitude, so the detection
    logits come out at +-1e5 and every sigmoid is exactly 0 or 1: no objectness spread for a threshold to cut, boxes the size of the anchors' squares.
    A trained network does not look like that.  This rescales the three detection convs (weights only, in place) so that each level's logits have unit
    spread around the synthetic biases — one eval forward of `x` (a couple of tiles) measures the spread.  Returns the per-level factors."""
    import torch
    head = model.headers[tag]
    was_training = model.training
    model.eval()
    with torch.no_grad():
        model(x)
        plan = list(model._eng().plans.values())[-1]
        stds = [float(d.float().std()) for d in plan.det_views()]
        for conv, s in zip(head.m, stds):
            conv.weight.div_(max(s, 1e-6))              # in place under no_grad: bumps the version counter, the eval plans re-pack
    model.train(was_training)
    return stds


def dense_conf_thres(preds_tile, survivors=1024):
    """confidence threshold at which one tile's decoded predictions (N, 5 + nc + 1) keep ~`survivors` candidates that also pass the 2-pixel size filter of
    nms_per_image (utils_general.py:332): the NMS load of a dense histology tile (10^3 - 10^4 nuclei) on a random-init network"""
    import torch
    ok = (preds_tile[:, 2] >= 2) & (preds_tile[:, 3] >= 2)
    obj = preds_tile[ok, 4].float()
    if obj.numel() <= survivors:
        return float(obj.min()) * 0.5 if obj.numel() else 0.0
    return float(torch.kthvalue(obj, obj.numel() - survivors).values)


def mask_state_dict(module, seed=0):
    """synth_state_dict with the detection convs damped, so that matched cells predict (almost) their anchor box: IoU >= 0.8 with the
    anchor-shaped truths of synth_mask_targets — otherwise nothing would reach the mask head."""
    sd = synth_state_dict(shapes_of(module), seed=seed)
    for k in list(sd):
        if k.startswith('headers.det.m.'):
            sd[k] = sd[k] * (0.02 if k.endswith('weight') else 0.0)
    return sd


def shapes_of(module):
    return {k: (tuple(v.shape), v.dtype) for k, v in module.state_dict().items()}


def synth_images(batch, size, seed=0):
    """Uniform [0,1) RGB tiles."""
    g = torch.Generator(device='cpu')
    g.manual_seed(1000 + seed)
    return torch.rand((batch, 3, size, size), generator=g, dtype=torch.float32)


def synth_targets(batch, size, nc, nmin=50, nmax=400, seed=1, task='det', normalize=True, masks=False):
    """Nuclei-like targets in the reference's batch schema (metayolo/datasets.py:462-519).

    boxes are xyxy, normalised to 0..1 when `normalize` (training convention,
    metayolo/datasets.py:496-497); labels are int64 in 1..nc.  With `masks`, every object also gets a (28, 28)
    float mask in its box frame (an ellipse of random fill; about one in ten is empty), as the mask branch consumes them.
    """
    g = torch.Generator(device='cpu')
    g.manual_seed(2000 + seed)
    targets = []
    for i in range(batch):
        n = int(torch.randint(nmin, nmax + 1, (1,), generator=g))
        cxy = 0.02 + 0.96 * torch.rand((n, 2), generator=g)
        wh = 0.015 + 0.045 * torch.rand((n, 2), generator=g)
        boxes = torch.cat([cxy - wh / 2, cxy + wh / 2], 1).clamp_(0.0, 1.0)
        if not normalize:
            boxes = boxes * size
        labels = torch.randint(1, nc + 1, (n,), generator=g, dtype=torch.int64)
        ann = {'size': torch.tensor([size, size], dtype=torch.int64),
               'boxes': boxes.float(), 'labels': labels}
        if masks:
            yy, xx = torch.meshgrid(torch.linspace(-1, 1, 28), torch.linspace(-1, 1, 28), indexing='ij')
            rad = 0.5 + 0.5 * torch.rand((n, 1, 1), generator=g)
            m = ((xx[None] ** 2 + yy[None] ** 2) <= rad ** 2).float()
            m[torch.rand(n, generator=g) < 0.1] = 0.0
            ann['masks'] = m
        targets.append({'image_id': torch.tensor([i], dtype=torch.int64),
                        'size': torch.tensor([size, size], dtype=torch.int64),
                        'anns': {task: [ann]}})
    return tuple(targets)


def synth_mask_targets(batch, size, nc, per_image=6, seed=4, task='det'):
    """Targets for exercising the mask branch: boxes that sit on a P3 / P4 cell centre with (almost) an anchor's shape, so that with
    near-zero box logits the predicted box of the matched cell overlaps its truth with IoU >= 0.8 (the reference only sends such
    proposals through the mask head, yolo_head.py:255-258); (28, 28) elliptic masks, one per image left empty."""
    g = torch.Generator(device='cpu')
    g.manual_seed(4000 + seed)
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, 28), torch.linspace(-1, 1, 28), indexing='ij')
    targets = []
    for i in range(batch):
        boxes = []
        for _ in range(per_image):
            lvl = int(torch.randint(0, 2, (1,), generator=g))
            stride = (8, 16)[lvl]
            aw, ah = ANCHORS_P5[lvl][2 * int(torch.randint(0, 3, (1,), generator=g)):][:2]
            n = size // stride
            gi, gj = (int(v) for v in torch.randint(1, n - 1, (2,), generator=g))
            jit = 1.0 + 0.06 * (torch.rand(2, generator=g) - 0.5)
            cx, cy, w, h = (gi + 0.5) * stride, (gj + 0.5) * stride, aw * float(jit[0]), ah * float(jit[1])
            boxes.append([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2])
        boxes = (torch.tensor(boxes) / size).clamp_(0.0, 1.0)
        labels = torch.randint(1, nc + 1, (per_image,), generator=g, dtype=torch.int64)
        rad = 0.5 + 0.5 * torch.rand((per_image, 1, 1), generator=g)
        m = ((xx[None] ** 2 + yy[None] ** 2) <= rad ** 2).float()
        m[0] = 0.0
        ann = {'size': torch.tensor([size, size], dtype=torch.int64), 'boxes': boxes.float(), 'labels': labels, 'masks': m}
        targets.append({'image_id': torch.tensor([i], dtype=torch.int64), 'size': torch.tensor([size, size], dtype=torch.int64),
                        'anns': {task: [ann]}})
    return tuple(targets)


def synth_nms_preds(batch, survivors, nc, size=1024, seed=2, pitch=12.0, extra=2000):
    """NMS stress input: `survivors` overlapping boxes on a jittered lattice
    with obj in (0.15, 1) plus `extra` background rows with obj < 0.15, shuffled.
    Returns preds (batch, survivors+extra, 5+nc+1) in the layout nms_per_image consumes
    (xywh px, obj, cls..., level id)."""
    g = torch.Generator(device='cpu')
    g.manual_seed(3000 + seed)
    n = survivors + extra
    out = torch.zeros((batch, n, 5 + nc + 1), dtype=torch.float32)
    side = max(int(survivors ** 0.5 + 0.999), 1)
    for b in range(batch):
        idx = torch.arange(survivors)
        cx = ((idx % side).float() + 0.5) * pitch + (torch.rand(survivors, generator=g) - 0.5) * pitch
        cy = ((idx // side).float() + 0.5) * pitch + (torch.rand(survivors, generator=g) - 0.5) * pitch
        wh = 14.0 + 14.0 * torch.rand((survivors, 2), generator=g)
        obj = 0.15 + 0.85 * torch.rand(survivors, generator=g)
        fg = torch.cat([cx[:, None], cy[:, None], wh, obj[:, None]], 1)
        bg = torch.cat([torch.rand((extra, 2), generator=g) * size,
                        4.0 + 30.0 * torch.rand((extra, 2), generator=g),
                        0.149 * torch.rand((extra, 1), generator=g)], 1)
        rows = torch.cat([fg, bg], 0)
        cls = torch.rand((n, nc), generator=g)
        lvl = torch.randint(0, 3, (n, 1), generator=g).float()
        rows = torch.cat([rows, cls, lvl], 1)
        perm = torch.randperm(n, generator=g)
        out[b] = rows[perm]
    return out
