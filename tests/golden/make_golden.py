#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE's own code.

Runs only in the build container (it reads /root/reference); the GPU box and the
test-suite only ever see the .npz files this script writes.  Usage:

    python tests/golden/make_golden.py

How the reference is imported: `metayolo/__init__.py` imports cv2 and
star-imports every sub-package, so the package object is pre-seeded with the three
names the hot-path modules take from it (LOGGER, check_version, load_cfg) and the
reference's own `metayolo/models/*.py` files are then imported unmodified from
/root/reference.  torchvision / torch_scatter are not installed; they are only
imported (never called) on the det-only path, except torchvision.ops.nms and
remove_small_boxes inside nms_per_image, for which the build's documented restatement
(oracle/nms_ref.py) is plugged in.  Consequently:

  * conv / BN / SiLU / C3 / SPPF / FPN stage outputs, det logits, decode, matcher,
    DetLoss and every gradient in these files are computed by the reference's code;
  * kept-box order in the `outputs_*` entries is computed by the reference's
    compute_outputs() on top of the oracle's NMS (torchvision boundary: unpinned).
"""
import importlib
import logging
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, ROOT)

from hd_yolo_amd import synth  # noqa: E402
from oracle import mask_ref, nms_ref     # noqa: E402


def install_shims():
    pkg = types.ModuleType('metayolo')
    pkg.__path__ = [os.path.join(REF, 'metayolo')]
    pkg.LOGGER = logging.getLogger('yolov5-ref')
    pkg.LOGGER.setLevel(logging.WARNING)

    def check_version(current='0.0.0', minimum='0.0.0', *a, **k):
        def t(v):
            return tuple(int(''.join(c for c in p if c.isdigit()) or 0) for p in v.split('+')[0].split('.')[:3])
        return t(current) >= t(minimum)

    def load_cfg(cfg):
        if isinstance(cfg, dict):
            return cfg
        import yaml
        with open(cfg) as f:
            return yaml.safe_load(f)

    pkg.check_version, pkg.load_cfg = check_version, load_cfg
    from pathlib import Path
    pkg.ROOT, pkg.check_python = Path(REF) / 'metayolo', (lambda *a, **k: None)       # engines/general.py takes these too
    sys.modules['metayolo'] = pkg
    eng = types.ModuleType('metayolo.engines')              # engines/__init__.py star-imports the training loop: skipped
    eng.__path__ = [os.path.join(REF, 'metayolo', 'engines')]
    sys.modules['metayolo.engines'] = eng
    if 'cv2' not in sys.modules:            # engines/general.py imports cv2 at module level and aliases three GUI functions
        cv2 = types.ModuleType('cv2')
        cv2.imshow = cv2.imread = cv2.imwrite = cv2.imdecode = cv2.imencode = (lambda *a, **k: None)
        cv2.IMREAD_COLOR = 1
        cv2.setNumThreads = lambda n: None
        sys.modules['cv2'] = cv2

    tv = types.ModuleType('torchvision')
    ops = types.ModuleType('torchvision.ops')
    misc = types.ModuleType('torchvision.ops.misc')

    def nms(boxes, scores, iou_threshold):
        keep = nms_ref.nms_numpy(boxes.detach().cpu().numpy().astype(np.float32),
                                 scores.detach().cpu().numpy().astype(np.float32), float(iou_threshold))
        return torch.from_numpy(keep).to(boxes.device)

    def remove_small_boxes(boxes, min_size):
        ws, hs = boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]
        return torch.where((ws >= min_size) & (hs >= min_size))[0]

    def _absent(*a, **k):
        raise RuntimeError('mask branch is out of scope for the goldens')

    class FrozenBatchNorm2d(nn.Module):
        pass

    # mask branch: the oracle's restatements of the absent third-party pieces (oracle/mask_ref.py; boundary unpinned)
    ops.nms, ops.remove_small_boxes, ops.roi_align = nms, remove_small_boxes, mask_ref.roi_align
    ops.FrozenBatchNorm2d = misc.FrozenBatchNorm2d = FrozenBatchNorm2d
    ops.misc = misc
    tv.ops = ops
    models = types.ModuleType('torchvision.models')
    det = types.ModuleType('torchvision.models.detection')
    mr = types.ModuleType('torchvision.models.detection.mask_rcnn')
    mr.MaskRCNNHeads, mr.MaskRCNNPredictor = mask_ref.MaskRCNNHeads, mask_ref.MaskRCNNPredictor
    tv.models, models.detection, det.mask_rcnn = models, det, mr
    for name, m in [('torchvision', tv), ('torchvision.ops', ops), ('torchvision.ops.misc', misc),
                    ('torchvision.models', models), ('torchvision.models.detection', det),
                    ('torchvision.models.detection.mask_rcnn', mr)]:
        sys.modules[name] = m
    ts = types.ModuleType('torch_scatter')
    ts.scatter_max = mask_ref.scatter_max
    sys.modules['torch_scatter'] = ts


def ref_model(variant, nc, hyp, masks=-1):
    yolo = importlib.import_module('metayolo.models.yolo')
    cfg = synth.make_cfg(variant, nc)
    cfg['headers'][0][3][3] = masks
    model = yolo.Model(cfg, hyp)
    sd = synth.synth_state_dict(synth.shapes_of(model), seed=0)
    missing = model.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys
    return model


def npf(t):
    return t.detach().cpu().numpy()


def gen_stages(tag, variant, nc, batch, size, full=True):
    """Eval-mode per-stage activations, det logits, decoded preds, final outputs."""
    hyp = synth.make_hyp(conf_thres=0.05)
    model = ref_model(variant, nc, hyp).eval()
    x = synth.synth_images(batch, size, seed=7)
    out = {'meta': np.array([batch, size, nc]), 'conf_thres': np.array(0.05)}
    with torch.no_grad():
        feats = model.backbone(x)
        for k, v in feats.items():
            out[f'backbone_{k}'] = npf(v)
        # stage-by-stage backbone (every layer, for localisation of a mismatch)
        h = x
        for i, m in enumerate(model.backbone):
            h = m(h)
            if full:
                out[f'stage_{i}'] = npf(h)
        neck = model.neck(dict(feats))
        for k, v in neck.items():
            out[f'neck_{k}'] = npf(v)
        head = model.headers['det']
        xs = [neck[j] for j in head.f]
        dets = []
        for i, conv in enumerate(head.m):
            f = conv(xs[i])
            bs, _, ny, nx = f.shape
            dets.append(f.view(bs, head.na, head.no, ny, nx).permute(0, 1, 3, 4, 2).contiguous())
            out[f'det_{i}'] = npf(dets[-1])
        preds = head.compute_proposals(dets)
        for i, p in enumerate(preds):
            out[f'pred_{i}'] = npf(p)
        _, outputs = model(x)
        for b, o in enumerate(outputs):
            out[f'out_{b}_boxes'] = npf(o['det']['boxes'])
            out[f'out_{b}_scores'] = npf(o['det']['scores'])
            out[f'out_{b}_labels'] = npf(o['det']['labels'])
        # fuse() equivalence: eval outputs after conv+bn folding
        if full:
            model.fuse()
            feats = model.neck(model.backbone(x))
            for k, v in feats.items():
                out[f'fused_neck_{k}'] = npf(v)
    np.savez_compressed(os.path.join(HERE, f'stages_{tag}.npz'), **out)
    print('wrote', f'stages_{tag}.npz', {k: v.shape for k, v in list(out.items())[:4]})


def gen_eval_compact(tag, variant, nc, batch, size, conf=0.05):
    """Full-size eval fixtures kept small: final detections per image, and per detection level the (sum, abs-sum, l2) of the logits plus
    a thin strip of them (first image, first anchor, four grid rows) to localise a mismatch."""
    hyp = synth.make_hyp(conf_thres=conf)
    model = ref_model(variant, nc, hyp).eval()
    x = synth.synth_images(batch, size, seed=7)
    out = {'meta': np.array([batch, size, nc]), 'conf_thres': np.array(conf)}
    with torch.no_grad():
        neck = model.neck(dict(model.backbone(x)))
        head = model.headers['det']
        for i, conv in enumerate(head.m):
            f = conv(neck[head.f[i]])
            bs, _, ny, nx = f.shape
            d = f.view(bs, head.na, head.no, ny, nx).permute(0, 1, 3, 4, 2).contiguous()
            d64 = d.double()
            out[f'det_{i}_sums'] = np.array([d64.sum().item(), d64.abs().sum().item(), d64.pow(2).sum().sqrt().item()])
            out[f'det_{i}_strip'] = npf(d[:1, :1, :4])
        _, outputs = model(x)
        for b, o in enumerate(outputs):
            out[f'out_{b}_boxes'] = npf(o['det']['boxes'])
            out[f'out_{b}_scores'] = npf(o['det']['scores'])
            out[f'out_{b}_labels'] = npf(o['det']['labels'])
    np.savez_compressed(os.path.join(HERE, f'eval_{tag}.npz'), **out)
    print('wrote', f'eval_{tag}.npz', [len(out[f'out_{b}_boxes']) for b in range(batch)])


def gen_eval_calibrated(tag, variant, nc, batch, size, conf=0.05):
    """bf16 detection-level fixture whose scores SEPARATE.  The random-init network's detection logits are flat (yolov5s) or
    saturated (yolov5l): which 300 of 25 200 candidates survive was decided below bf16 resolution, so a recall against those fixtures said nothing
    about the bf16 kernels.  Here the three detection convs of the REFERENCE model are rescaled — on the reference, on the CPU, in fp32 — so that each
    level's logits have unit spread (the calibration bench.py's C4 leg applies, synth.calibrate_det_logits); the factors travel in the fixture and the
    test applies the same ones, so both sides run identical weights.  Everything stored is the reference's own output."""
    hyp = synth.make_hyp(conf_thres=conf)
    model = ref_model(variant, nc, hyp).eval()
    x = synth.synth_images(batch, size, seed=7)
    out = {'meta': np.array([batch, size, nc]), 'conf_thres': np.array(conf)}
    with torch.no_grad():
        neck = model.neck(dict(model.backbone(x)))
        head = model.headers['det']
        scales = []
        for i, conv in enumerate(head.m):
            sd = float(conv(neck[head.f[i]]).float().std())
            scales.append(max(sd, 1e-6))
            conv.weight.div_(scales[-1])
        out['det_scales'] = np.array(scales, dtype=np.float64)
        for i, conv in enumerate(head.m):
            f = conv(neck[head.f[i]])
            bs, _, ny, nx = f.shape
            d = f.view(bs, head.na, head.no, ny, nx).permute(0, 1, 3, 4, 2).contiguous()
            d64 = d.double()
            out[f'det_{i}_sums'] = np.array([d64.sum().item(), d64.abs().sum().item(), d64.pow(2).sum().sqrt().item()])
            out[f'det_{i}_strip'] = npf(d[:1, :1, :4])
        _, outputs = model(x)
        for b, o in enumerate(outputs):
            out[f'out_{b}_boxes'] = npf(o['det']['boxes'])
            out[f'out_{b}_scores'] = npf(o['det']['scores'])
            out[f'out_{b}_labels'] = npf(o['det']['labels'])
    np.savez_compressed(os.path.join(HERE, f'eval_cal_{tag}.npz'), **out)
    sc = np.concatenate([out[f'out_{b}_scores'] for b in range(batch)])
    print('wrote', f'eval_cal_{tag}.npz', [len(out[f'out_{b}_boxes']) for b in range(batch)], 'scores min / median / max', sc.min(), np.median(sc), sc.max(),
          'scales', scales)


def gen_trajectory(tag, variant, nc, batch, size, steps, nmin, nmax, lr=0.01):
    """Loss trajectory of the reference over `steps` optimizer steps on one fixed batch: torch.optim.SGD(momentum 0.937, nesterov) over the three
    parameter groups of train.py:208-233 with a constant learning rate — the loop body of train.py:455-472 without the loader.  Pins that the HIP
    path TRAINS like the reference (fp32 step by step) and bounds what bf16 storage does to the trajectory (tests/test_gpu_model.py).
    The loop is chaotic (train-mode BatchNorm over 4 tiles, random weights): the fixture therefore also holds the reference's OWN deviation from
    itself — the same loop with 8 host threads instead of 1 (another fp32 summation order) and with every weight perturbed by 1e-6 relative — which is
    the band any faithful implementation can be held to beyond the first few steps."""
    hyp = synth.make_hyp()

    def run(threads, perturb=0.0):
        torch.set_num_threads(threads)          # 1: deterministic objectness scatter (see gen_train)
        model = ref_model(variant, nc, hyp).train()
        if perturb:
            gen = torch.Generator().manual_seed(123)
            with torch.no_grad():
                for q in model.parameters():
                    q.mul_(1 + perturb * torch.randn(q.shape, generator=gen))
        x = synth.synth_images(batch, size, seed=11)
        targets = synth.synth_targets(batch, size, nc, nmin=nmin, nmax=nmax, seed=5)
        g_bn, g_w, g_b = [], [], []
        for m in model.modules():
            if hasattr(m, 'bias') and isinstance(m.bias, nn.Parameter):
                g_b.append(m.bias)
            if isinstance(m, nn.BatchNorm2d):
                g_bn.append(m.weight)
            elif hasattr(m, 'weight') and isinstance(m.weight, nn.Parameter):
                g_w.append(m.weight)
        opt = torch.optim.SGD(g_bn, lr=lr, momentum=hyp['momentum'], nesterov=True)
        opt.add_param_group({'params': g_w, 'weight_decay': hyp['weight_decay']})
        opt.add_param_group({'params': g_b})
        losses = []
        for _ in range(steps):
            out, _ = model(x, targets, compute_masks=True)
            loss = out['det']['det_loss']
            loss.backward()
            opt.step()
            opt.zero_grad(set_to_none=True)
            losses.append(float(loss.detach()))
        return np.array(losses, dtype=np.float64)

    losses, l8, lp = run(1), run(8), run(1, 1e-6)
    torch.set_num_threads(8)
    np.savez_compressed(os.path.join(HERE, f'trajectory_{tag}.npz'), meta=np.array([batch, size, nc, nmin, nmax, steps]), lr=np.array(lr),
                        weight_decay=np.array(hyp['weight_decay']), momentum=np.array(hyp['momentum']), losses=losses, losses_8_threads=l8,
                        losses_perturbed_1e6=lp)
    print('wrote', f'trajectory_{tag}.npz', [round(v, 4) for v in losses], 'own deviation: 8 threads', float((np.abs(l8 - losses) / losses).max()),
          'weights * (1 + 1e-6 n)', float((np.abs(lp - losses) / losses).max()))


GRAD_KEYS = ['backbone.0.conv.weight', 'backbone.1.conv.weight', 'backbone.2.m.0.cv2.conv.weight',
             'backbone.2.cv3.conv.weight', 'backbone.2.cv1.bn.weight', 'backbone.2.cv1.bn.bias',
             'backbone.9.cv2.conv.weight', 'neck.3.cv3.conv.weight', 'neck.8.conv.weight',
             'headers.det.m.0.weight', 'headers.det.m.0.bias', 'headers.det.m.2.weight',
             'headers.det.m.3.weight', 'headers.det.m.3.bias', 'neck.16.conv.weight']       # (keys absent from a graph are skipped)
STAT_KEYS = ['backbone.0.bn.running_mean', 'backbone.0.bn.running_var',
             'backbone.4.m.1.cv2.bn.running_mean', 'backbone.4.m.1.cv2.bn.running_var',
             'neck.13.cv3.bn.running_mean', 'neck.13.cv3.bn.running_var',
             'neck.20.cv3.bn.running_mean', 'neck.20.cv3.bn.running_var']


def gen_train(tag, variant, nc, batch, size, nmin, nmax, empty_first=False):
    """Train-mode: loss dict, BN running stats after one forward, gradients."""
    hyp = synth.make_hyp()
    model = ref_model(variant, nc, hyp).train()
    x = synth.synth_images(batch, size, seed=11)
    targets = synth.synth_targets(batch, size, nc, nmin=nmin, nmax=nmax, seed=5)
    if empty_first:                           # a tile without nuclei: empty boxes / labels (datasets.py:462-519)
        a = targets[0]['anns']['det'][0]
        a['boxes'], a['labels'] = a['boxes'][:0], a['labels'][:0]
    # Dense targets (50-400 per 640x640 tile) put several matches into the same (anchor, cell); `tobj[b, a, gj, gi] = iou` (loss.py:217)
    # then keeps whichever write its parallel index_put_ made last: with 8 threads the obj term moved by 1e-4 and the stem gradient by
    # 2e-3 between runs.  One thread makes it the last match in target order — the semantics the oracle and csrc/loss.hip implement.
    threads = torch.get_num_threads()
    if size >= 640:
        torch.set_num_threads(1)
    losses, _ = model(x, targets, compute_masks=True)
    loss = losses['det']['det_loss'] + losses['det']['mask_loss']
    loss.backward()
    torch.set_num_threads(threads)
    out = {'meta': np.array([batch, size, nc, nmin, nmax]),
           'loss': npf(losses['det']['det_loss'])}
    for k, v in losses['det']['loss_items'].items():
        out[f'loss_{k}'] = npf(v)
    sd = model.state_dict()
    for k in STAT_KEYS:
        if k in sd:
            out['stat:' + k] = npf(sd[k])
    params = dict(model.named_parameters())
    for k in GRAD_KEYS:
        if k in params and params[k].numel() <= 40000:      # keep fixtures small; gradsum below covers the rest
            out['grad:' + k] = npf(params[k].grad)
    # one (sum, abs-sum, l2) triple per parameter: catches a wrong gradient anywhere
    names, sums = [], []
    for k, p in params.items():
        g = p.grad
        if g is None:
            continue
        names.append(k)
        g64 = g.double()
        sums.append([g64.sum().item(), g64.abs().sum().item(), g64.pow(2).sum().sqrt().item()])
    out['gradsum_names'] = np.array(names)
    out['gradsum'] = np.array(sums, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, f'train_{tag}.npz'), **out)
    print('wrote', f'train_{tag}.npz', 'loss', out['loss'], {k: float(out[k].reshape(-1)[0]) for k in out if k.startswith('loss_')})


def gen_decode():
    """Decode-only vectors: default grid sizes and a non-default one (forces _make_grid)."""
    hyp = synth.make_hyp()
    model = ref_model('n', 2, hyp).eval()
    head = model.headers['det']
    g = torch.Generator().manual_seed(99)
    out = {}
    for tag, sizes in [('default', [(80, 80), (40, 40), (20, 20)]), ('odd', [(12, 20), (6, 10), (3, 5)])]:
        dets = [torch.randn((2, head.na, ny, nx, head.no), generator=g) * 2.0 for ny, nx in sizes]
        with torch.no_grad():
            preds = head.compute_proposals(dets)
        for i, (d, p) in enumerate(zip(dets, preds)):
            if tag == 'default':     # keep the fixture small: one image, a strip of rows
                d, p = d[:1, :, :4], p[:1, :, :4]
            out[f'{tag}_det_{i}'] = npf(d)
            out[f'{tag}_pred_{i}'] = npf(p)
    np.savez_compressed(os.path.join(HERE, 'decode.npz'), **out)
    print('wrote decode.npz')


def gen_outputs():
    """compute_outputs(): score/label logic incl. the -100 label and multi_label=True.
    NMS inside is the oracle's (see module docstring)."""
    out = {}
    for tag, ml in [('single', False), ('multi', True)]:
        hyp = synth.make_hyp(conf_thres=0.15, multi_label=ml)
        model = ref_model('n', 3, hyp).eval()
        head = model.headers['det']
        g = torch.Generator().manual_seed(123)
        # logits centred so that a good share of candidates pass obj > conf and overlap
        dets = [torch.randn((2, head.na, ny, nx, head.no), generator=g) * 1.5 for ny, nx in [(8, 8), (4, 4), (2, 2)]]
        with torch.no_grad():
            preds = head.compute_proposals([d.clone() for d in dets])
            res = head.compute_outputs(preds, [], compute_masks=False)
        for i, d in enumerate(dets):
            out[f'{tag}_det_{i}'] = npf(d)
        for b, r in enumerate(res):
            out[f'{tag}_out_{b}_boxes'] = npf(r['boxes'])
            out[f'{tag}_out_{b}_scores'] = npf(r['scores'])
            out[f'{tag}_out_{b}_labels'] = npf(r['labels'])
    np.savez_compressed(os.path.join(HERE, 'outputs.npz'), **out)
    print('wrote outputs.npz')


def gen_seg():
    """hnet's semantic-segmentation header from the reference's OWN classes: PanopticFeatureConnector
    (hnet/segmentation/utils_seg.py) unmodified, and PanopticSeg (hnet/segmentation/panoptic_seg.py) with the two names it cannot
    resolve supplied: `SoftDiceLoss` (defined nowhere upstream: oracle/seg_ref.py's restatement from the repository's dice) and
    torchvision's roi_align on its identity case (a whole-map roi at the map's own resolution with aligned=True samples every bin
    once, at the pixel centre).  `hnet/__init__.py` imports timm / mmcv-style modules, so the package object is pre-seeded."""
    import contextlib
    import io
    from collections import OrderedDict
    from oracle import seg_ref
    pkg = types.ModuleType('hnet')
    pkg.__path__ = [os.path.join(REF, 'hnet')]
    sys.modules['hnet'] = pkg
    segpkg = types.ModuleType('hnet.segmentation')
    segpkg.__path__ = [os.path.join(REF, 'hnet', 'segmentation')]
    sys.modules['hnet.segmentation'] = segpkg
    tv_ops = sys.modules['torchvision.ops']

    def roi_align_identity(f, rois, o_size, aligned=False, **k):
        assert aligned and tuple(o_size) == tuple(f.shape[2:])
        for r in rois:
            assert r.shape == (1, 4) and [float(v) for v in r[0]] == [0.0, 0.0, float(f.shape[3]), float(f.shape[2])], r
        return f
    saved = tv_ops.roi_align
    tv_ops.roi_align = roi_align_identity
    boxes_mod = types.ModuleType('torchvision.ops.boxes')     # sliding_window_scanner clips its windows to the image (hnet/utils.py:60)

    def clip_boxes_to_image(boxes, size):
        h, w = size
        return torch.stack([boxes[:, 0].clamp(0, w), boxes[:, 1].clamp(0, h), boxes[:, 2].clamp(0, w), boxes[:, 3].clamp(0, h)], 1)
    boxes_mod.clip_boxes_to_image = clip_boxes_to_image
    tv_ops.boxes = boxes_mod
    useg = importlib.import_module('hnet.segmentation.utils_seg')
    pseg = importlib.import_module('hnet.segmentation.panoptic_seg')

    class SoftDiceLoss(nn.Module):
        def __init__(self, class_weight=None):
            super().__init__()
            self.class_weight = class_weight

        def forward(self, probs, masks):
            return seg_ref.soft_dice_criterion(probs, masks, self.class_weight)
    pseg.SoftDiceLoss = SoftDiceLoss
    out = {}
    g = torch.Generator().manual_seed(77)
    # ---- connector alone: 4 levels (the reference's pyramid depth), widths as a yolov5n6-like pyramid, 32-channel ladders (one channel per group)
    chans, sizes = [32, 64, 96, 128], [(24, 16), (12, 8), (6, 4), (3, 2)]
    names = OrderedDict((str(k), str(k)) for k in (23, 26, 29, 32))
    con = useg.PanopticFeatureConnector(chans, 32, names)
    for k, v in con.state_dict().items():
        v.copy_(torch.randn(v.shape, generator=g) * (0.08 if v.dim() == 4 else 0.5) + (1.0 if k.endswith('1.weight') or '.weight' in k and v.dim() == 1 else 0.0))
    feats = OrderedDict((n, torch.randn((2, c, h, w), generator=g, requires_grad=True)) for n, c, (h, w) in zip(names, chans, sizes))
    y = con(feats)['0']
    wsum = torch.randn(y.shape, generator=g)
    (y * wsum).sum().backward()
    out['con_out'], out['con_wsum'] = npf(y), npf(wsum)
    for n, f in feats.items():
        out[f'con_in_{n}'], out[f'con_din_{n}'] = npf(f), npf(f.grad)
    for k, v in con.state_dict().items():
        out[f'con_p_{k}'] = npf(v)
    for k, v in con.named_parameters():
        out[f'con_g_{k}'] = npf(v.grad)
    # ---- PanopticSeg: 3 levels at 1/8, 1/16, 1/32 of a 64 x 96 tile, 3 classes, scale_factor 8, class weights
    cfg = {'in_channels': 64, 'num_classes': 3, 'feature_maps': OrderedDict((str(k), str(k)) for k in (17, 20, 23)), 'scale_factor': 8,
           'resize_mode': 'bilinear', 'class_weight': [1.0, 2.0, 0.5], 'roi_size': None}
    head = pseg.PanopticSeg(cfg).train()
    for k, v in head.state_dict().items():
        v.copy_(torch.randn(v.shape, generator=g) * (0.08 if v.dim() == 4 else 0.5) + (1.0 if v.dim() == 1 and k.endswith('weight') else 0.0))
    H, W = 64, 96
    feats = OrderedDict((n, torch.randn((2, 64, H // s, W // s), generator=g, requires_grad=True)) for n, s in zip(cfg['feature_maps'], (8, 16, 32)))
    lab = torch.randint(0, 3, (2, H, W), generator=g)
    masks = F_one_hot(lab, 3)
    targets = [[{'roi': torch.tensor([0.0, 0.0, W, H]), 'masks': masks[i]}] for i in range(2)]
    with contextlib.redirect_stdout(io.StringIO()):                      # the reference prints a debug line per level (hnet/utils.py:152)
        res, losses = head(feats, (H, W), None, targets)
    losses['soft_iou_loss'].backward()
    out['seg_loss'] = npf(losses['soft_iou_loss'].reshape(1))
    out['seg_probs'] = npf(torch.cat(list(res)))
    out['seg_masks'] = npf(masks)
    for n, f in feats.items():
        out[f'seg_in_{n}'], out[f'seg_din_{n}'] = npf(f), npf(f.grad)
    for k, v in head.state_dict().items():
        out[f'seg_p_{k}'] = npf(v)
    for k, v in head.named_parameters():
        out[f'seg_g_{k}'] = npf(v.grad)
    head.eval()
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        res, _ = head(OrderedDict((n, f.detach()) for n, f in feats.items()), (H, W), None, None)
    out['seg_eval_probs'] = npf(torch.cat(list(res)))
    tv_ops.roi_align = saved
    for m in ('hnet', 'hnet.segmentation', 'hnet.segmentation.utils_seg', 'hnet.segmentation.panoptic_seg', 'hnet.utils'):
        sys.modules.pop(m, None)
    np.savez_compressed(os.path.join(HERE, 'seg.npz'), **out)
    print('wrote seg.npz')


def F_one_hot(lab, nc):
    return torch.nn.functional.one_hot(lab, nc).permute(0, 3, 1, 2).float().contiguous()


def gen_keys():
    """state_dict key list + shapes for the four variants (drop-in surface)."""
    out = {}
    for v, nc in [('n', 2), ('s', 8), ('m', 8), ('l', 8)]:
        model = ref_model(v, nc, synth.make_hyp())
        sd = model.state_dict()
        out[f'{v}_keys'] = np.array(list(sd.keys()))
        out[f'{v}_shapes'] = np.array([','.join(map(str, t.shape)) for t in sd.values()])
        out[f'{v}_nparams'] = np.array(sum(p.numel() for p in model.parameters()))
    np.savez_compressed(os.path.join(HERE, 'keys.npz'), **out)
    print('wrote keys.npz', {k: int(v) for k, v in out.items() if k.endswith('nparams')})


def synth_detections(seed=3):
    """A small seeded validation set: per image ground truth (boxes px, labels in {1,2,3,-100}) and detections = jittered
    copies of some truths + false positives, with a few wrong and a few -100 labels."""
    g = torch.Generator().manual_seed(seed)
    images = []
    for n_true in (0, 3, 12, 20, 7, 15):
        c = torch.rand((n_true, 2), generator=g) * 200 + 20
        wh = torch.rand((n_true, 2), generator=g) * 30 + 10
        tb = torch.cat([c - wh / 2, c + wh / 2], 1)
        tl = torch.randint(1, 4, (n_true,), generator=g)
        tl[torch.rand(n_true, generator=g) < 0.15] = -100
        hit = torch.rand(n_true, generator=g) < 0.8
        pb = tb[hit] + (torch.rand((int(hit.sum()), 4), generator=g) - 0.5) * 8
        pl = tl[hit].clone()
        wrong = torch.rand(len(pl), generator=g) < 0.2
        pl[wrong] = torch.randint(1, 4, (int(wrong.sum()),), generator=g)
        n_fp = int(torch.randint(0, 6, (1,), generator=g))
        c = torch.rand((n_fp, 2), generator=g) * 200 + 20
        wh = torch.rand((n_fp, 2), generator=g) * 30 + 10
        pb = torch.cat([pb, torch.cat([c - wh / 2, c + wh / 2], 1)])
        fl = torch.randint(1, 4, (n_fp,), generator=g)
        fl[torch.rand(n_fp, generator=g) < 0.3] = -100
        pl = torch.cat([pl, fl])
        ps = torch.rand(len(pl), generator=g)
        images.append(({'boxes': pb, 'scores': ps, 'labels': pl}, {'boxes': tb, 'labels': tl}))
    return images


def gen_f3():
    """APMeter, Ensemble.merge, scale_coords, checkpoint re-keying and header label re-ordering."""
    metrics = importlib.import_module('metayolo.models.metrics')
    yolo = importlib.import_module('metayolo.models.yolo')
    ug = importlib.import_module('metayolo.models.utils_general')
    general = importlib.import_module('metayolo.engines.general')
    out = {}
    # --- APMeter
    meter = metrics.APMeter({1: 'a', 2: 'b', 3: 'c'})
    for i, (o, t) in enumerate(synth_detections()):
        for k, v in o.items():
            out[f'ap_in_{i}_o_{k}'] = npf(v)
        for k, v in t.items():
            out[f'ap_in_{i}_t_{k}'] = npf(v)
        meter.add(o, t)
    # iouv as the reference's caller passes it (val_nuclei.py:56): the np.linspace default does not run on this torch
    for tag, kw in [('default', {}), ('noignore', {'ignore': []})]:
        st = meter.ap_per_class(iouv=torch.linspace(0.5, 0.95, 10), **kw)
        out[f'ap_{tag}_labels'] = np.array(st['labels'])
        out[f'ap_{tag}_counts'] = np.array(st['counts'])
        for k in ('py', 'ap', 'p', 'r', 'f1'):
            out[f'ap_{tag}_{k}'] = np.asarray(st[k], dtype=np.float64)
    # --- Ensemble.merge (NMS = the oracle's torchvision restatement)
    ens = yolo.Ensemble([], {'conf_thres': 0.3, 'iou_thres': 0.4, 'max_det': 12})
    parts = []
    for j, (o, _) in enumerate(synth_detections(seed=8)[2:5]):
        parts.append({'det': {k: v.clone() for k, v in o.items()}})
        for k, v in o.items():
            out[f'ens_in_{j}_{k}'] = npf(v)
    merged = ens.merge(parts)['det']
    for k, v in merged.items():
        out[f'ens_out_{k}'] = npf(v)
    # --- scale_coords
    g = torch.Generator().manual_seed(4)
    coords = torch.rand((9, 4), generator=g) * 640
    out['sc_in'] = npf(coords)
    out['sc_a'] = npf(ug.scale_coords((640, 640), coords.clone(), (480, 720)))
    out['sc_b'] = npf(ug.scale_coords((512, 640), coords.clone(), (1000, 900), ratio_pad=((0.5, 0.5), (12.0, 7.0))))
    # --- stock-YOLOv5 key conversion: keys only
    model = ref_model('n', 2, synth.make_hyp())
    nb, nn_ = len(model.backbone), len(model.neck)
    stock = {}
    for k, v in model.state_dict().items():
        part, idx, rest = k.split('.', 2)
        if part == 'backbone':
            stock[f'model.{idx}.{rest}'] = v
        elif part == 'neck':
            stock[f'model.{int(idx) + nb}.{rest}'] = v
        elif part == 'headers' and rest.startswith('m.'):
            stock[f'model.{nb + nn_}.{rest}'] = v
    conv = general.convert_yolo_weights(model, stock)
    out['conv_in_keys'] = np.array(list(stock.keys()))
    out['conv_out_keys'] = np.array(list(conv.keys()))
    # --- header label re-ordering: new class i <- old class label_map[i]
    model = ref_model('n', 3, synth.make_hyp())
    head = model.headers['det']
    for i, m in enumerate(head.m):
        out[f'lm_in_w{i}'], out[f'lm_in_b{i}'] = npf(m.weight), npf(m.bias)
    label_map = [2, 0, -1, 1]
    general.manipulate_header_label_order(head, label_map)
    out['lm_map'] = np.array(label_map)
    for i, m in enumerate(head.m):
        out[f'lm_out_w{i}'], out[f'lm_out_b{i}'] = npf(m.weight), npf(m.bias)
    np.savez_compressed(os.path.join(HERE, 'f3.npz'), **out)
    print('wrote f3.npz', out['ap_default_ap'].round(3).tolist(), out['ens_out_labels'].tolist(), out['conv_out_keys'][:2])


NMS_OPTION_CASES = {
    'default': {},
    'multi': {'multi_label': True},
    'agnostic': {'agnostic': True},
    'classes': {'classes': [0, 2]},
    'multi_agnostic_top5': {'multi_label': True, 'agnostic': True, 'max_det': 5},
    'apriori': {'labels': 'LABELS'},
}


def nms_option_inputs():
    preds = synth.synth_nms_preds(2, 400, 3, seed=21, extra=200).numpy()[:, :, :8].copy()
    g = np.random.default_rng(5)
    labels = [np.concatenate([g.integers(0, 3, (4, 1)).astype(np.float32), g.uniform(40, 300, (4, 2)).astype(np.float32),
                              g.uniform(10, 30, (4, 2)).astype(np.float32)], 1), np.zeros((0, 5), np.float32)]
    return preds, labels


def gen_nms_options():
    """non_max_suppression (utils_general.py:423-523) with every option, NMS step = the oracle's torchvision restatement."""
    ug = importlib.import_module('metayolo.models.utils_general')
    preds, labels = nms_option_inputs()
    out = {'preds': preds, 'labels_0': labels[0], 'labels_1': labels[1]}
    for tag, kw in NMS_OPTION_CASES.items():
        kw = dict(kw)
        if kw.get('labels') == 'LABELS':
            kw['labels'] = [torch.from_numpy(l) for l in labels]
        res = ug.non_max_suppression(torch.from_numpy(preds.copy()), conf_thres=0.2, iou_thres=0.5, **kw)
        for b, d in enumerate(res):
            out[f'{tag}_{b}'] = npf(d)
    np.savez_compressed(os.path.join(HERE, 'nms_options.npz'), **out)
    print('wrote nms_options.npz', {k: v.shape for k, v in out.items() if k.endswith('_0')})


def gen_masks():
    """the reference's Detect with masks=1, on the oracle's roi_align / Mask R-CNN head / scatter_max."""
    out = {}
    # ---- eval: detections with 28 x 28 masks
    hyp = synth.make_hyp(conf_thres=0.05)
    model = ref_model('n', 2, hyp, masks=1)
    model.load_state_dict(synth.mask_state_dict(model), strict=False)
    model.eval()
    x = synth.synth_images(2, 128, seed=7)
    # compute_outputs(compute_masks=True) itself does not run on this torch (yolo_head.py:348 indexes with the float result of
    # clamp(min=0.) -> IndexError), so its steps are called one by one: every call below is the reference's own function
    from torch.nn import functional as F
    ug = importlib.import_module('metayolo.models.utils_general')
    with torch.no_grad():
        head = model.headers['det']
        feats = model.neck(model.backbone(x))
        xs = [feats[j] for j in head.f]
        dets = []
        for i, conv in enumerate(head.m):
            f = conv(xs[i])
            bs, _, ny, nx = f.shape
            dets.append(f.view(bs, head.na, head.no, ny, nx).permute(0, 1, 3, 4, 2).contiguous())
        preds = head.compute_proposals(dets)
        mask_maps = [seg(xs[-i]) for i, seg in enumerate(head.seg, 1)][::-1]
        flat = torch.cat([F.pad(y.view(y.shape[0], -1, head.no), [0, 1], value=float(idx)) for idx, y in enumerate(preds)], 1)
        kept = ug.nms_per_image(flat, nc=head.nc, conf_thres=head.nms_params['conf_thres'], iou_thres=head.nms_params['iou_thres'],
                                max_det=int(head.nms_params['max_det']))
        proposals = torch.cat([F.pad(k['boxes'], [1, 0], value=float(i)) for i, k in enumerate(kept) if len(k['boxes'])])
        levels = torch.cat([k['extra'][:, 0] for k in kept if len(k['boxes'])])
        probs = head.seg_h(head.multiscale_roi_align(mask_maps, boxes=proposals, levels=levels)).sigmoid()
        _, outputs = model(x, compute_masks=False)
    out['eval_mask_probs'] = npf(probs)                       # (R, nc_masks, 28, 28), detections of image 0 then image 1
    out['eval_levels'] = npf(levels)
    for l, f in enumerate(mask_maps):
        out[f'eval_maskmap_{l}'] = npf(f)
    for b, o in enumerate(outputs):
        for k in ('boxes', 'scores', 'labels'):
            out[f'eval_{b}_{k}'] = npf(o['det'][k])
    # ---- train: det + mask loss, gradients
    model = ref_model('n', 2, synth.make_hyp(), masks=1)
    model.load_state_dict(synth.mask_state_dict(model), strict=False)
    model.train()
    x = synth.synth_images(2, 128, seed=11)
    targets = synth.synth_mask_targets(2, 128, 2, per_image=6, seed=4)
    losses, _ = model(x, targets, compute_masks=True)
    (losses['det']['det_loss'] + losses['det']['mask_loss']).backward()
    out['train_det_loss'], out['train_mask_loss'] = npf(losses['det']['det_loss']), npf(losses['det']['mask_loss'])
    params = dict(model.named_parameters())
    names, sums = [], []
    for k, p in params.items():
        if p.grad is None:
            continue
        names.append(k)
        g64 = p.grad.double()
        sums.append([g64.sum().item(), g64.abs().sum().item(), g64.pow(2).sum().sqrt().item()])
    out['gradsum_names'], out['gradsum'] = np.array(names), np.array(sums, dtype=np.float64)
    for k in ('headers.det.seg_h.maskrcnn_preds.mask_fcn_logits.weight', 'headers.det.seg_h.maskrcnn_preds.mask_fcn_logits.bias',
              'headers.det.seg_h.maskrcnn_heads.mask_fcn1.bias', 'headers.det.seg.0.bn.weight', 'headers.det.seg.2.bn.bias'):
        out['grad:' + k] = npf(params[k].grad)
    np.savez_compressed(os.path.join(HERE, 'masks.npz'), **out)
    print('wrote masks.npz', 'mask_loss', out['train_mask_loss'], 'eval dets', [out[f'eval_{b}_boxes'].shape for b in range(2)],
          'mask grad l2', float(out['gradsum'][names.index('headers.det.seg_h.maskrcnn_heads.mask_fcn1.weight'), 2]))


def gen_scale_img():
    """utils_torch.scale_img (test-time augmentation resize + pad) on a seeded batch, three (ratio, same_shape) settings."""
    ut = importlib.import_module('metayolo.models.utils_torch')
    g = torch.Generator().manual_seed(21)
    x = torch.rand((2, 3, 64, 96), generator=g)
    out = {'x': npf(x)}
    for i, (r, ss) in enumerate(((0.83, False), (0.67, True), (1.5, False))):
        out[f'y{i}'] = npf(ut.scale_img(x, r, ss, gs=32))
        out[f'arg{i}'] = np.array([r, float(ss)])
    np.savez_compressed(os.path.join(HERE, 'scale_img.npz'), **out)
    print('wrote scale_img.npz', [out[f'y{i}'].shape for i in range(3)])


def gen_confusion():
    """metrics.ConfusionMatrix.process_batch over three seeded images (jittered copies of the labels + clutter), and tp_fp()."""
    mt = importlib.import_module('metayolo.models.metrics')
    g = torch.Generator().manual_seed(33)
    nc = 4
    cm = mt.ConfusionMatrix(nc=nc, conf=0.25, iou_thres=0.45)
    out = {'nc': np.array(nc)}
    for i in range(3):
        m = 12 + 3 * i
        xy = torch.rand((m, 2), generator=g) * 400
        wh = torch.rand((m, 2), generator=g) * 60 + 10
        lab_boxes = torch.cat([xy, xy + wh], 1)
        labels = torch.cat([torch.randint(0, nc, (m, 1), generator=g).float(), lab_boxes], 1)
        jit = lab_boxes[: m - 3] + (torch.rand((m - 3, 4), generator=g) - 0.5) * 14
        clutter_xy = torch.rand((5, 2), generator=g) * 400
        clutter = torch.cat([clutter_xy, clutter_xy + 30], 1)
        boxes = torch.cat([jit, clutter, lab_boxes[:2] + 1.0])                 # two labels get a second, near-duplicate detection
        conf = torch.rand((len(boxes), 1), generator=g)
        cls = torch.cat([labels[: m - 3, :1], torch.randint(0, nc, (7, 1), generator=g).float()])
        flip = torch.rand((len(boxes), 1), generator=g) < 0.25
        cls = torch.where(flip, (cls + 1) % nc, cls)
        det = torch.cat([boxes, conf, cls], 1)
        cm.process_batch(det, labels)
        out[f'det{i}'], out[f'lab{i}'] = npf(det), npf(labels)
    out['matrix'] = np.array(cm.matrix)
    tp, fp = cm.tp_fp()
    out['tp'], out['fp'] = np.array(tp), np.array(fp)
    np.savez_compressed(os.path.join(HERE, 'confusion.npz'), **out)
    print('wrote confusion.npz', out['matrix'].sum(), out['tp'], out['fp'])


def main():
    assert os.path.isdir(REF), 'the reference is only mounted in the build container'
    torch.set_num_threads(8)
    install_shims()
    if sys.argv[1:] == ['ragged']:
        gen_train('n_64_ragged', 'n', 2, 2, 64, 3, 8, empty_first=True)
        return
    if sys.argv[1:] == ['confusion']:
        gen_confusion()
        return
    if sys.argv[1:] == ['scale_img']:
        gen_scale_img()
        return
    if sys.argv[1:] == ['fullsize']:            # BASELINE config C1 exactly (yolov5n, 2 classes, batch 4, 640x640) and yolov5s at 640x640
        gen_train('c1_640', 'n', 2, 4, 640, 50, 400)
        gen_eval_compact('c1_640', 'n', 2, 4, 640)
        gen_eval_compact('s_640', 's', 8, 2, 640)
        return
    if sys.argv[1:] == ['calibrated']:         # round 6: bf16 detection-level fixtures whose scores separate; a 30-step loss trajectory
        gen_eval_calibrated('s_640', 's', 8, 2, 640)
        gen_eval_calibrated('l_256', 'l', 8, 1, 256)
        gen_trajectory('n_128', 'n', 2, 4, 128, 30, 4, 12)
        return
    if sys.argv[1:] == ['sizes']:              # round 6: BASELINE configs[2] / [3] at their own tile geometry (per-GPU batch cut to what a CPU finishes in a minute)
        gen_train('m_640', 'm', 8, 2, 640, 50, 400)
        gen_eval_calibrated('l_1024', 'l', 8, 1, 1024)
        return
    if sys.argv[1:] == ['trajectory']:
        gen_trajectory('n_128', 'n', 2, 4, 128, 30, 4, 12)
        return
    if sys.argv[1:] == ['seg']:
        gen_seg()
        return
    if sys.argv[1:] == ['m']:                  # BASELINE config C3's variant: channel widths 48 / 96 / 192 / 384 / 768, C3 depth 2 / 4 / 6 / 2
        gen_train('m_128', 'm', 8, 2, 128, 10, 30)
        gen_eval_compact('m_128', 'm', 8, 2, 128)
        return
    if sys.argv[1:] == ['p6']:                 # only the P6 fixtures (the others are unchanged by construction)
        gen_stages('n6_128', 'n6', 3, 2, 128, full=False)
        gen_train('n6_128', 'n6', 3, 2, 128, 4, 12)
        return
    gen_keys()
    gen_decode()
    gen_outputs()
    gen_stages('n_64', 'n', 2, 2, 64)
    gen_stages('s_128', 's', 8, 1, 128, full=False)
    gen_train('n_64', 'n', 2, 2, 64, 3, 8)
    gen_train('s_128', 's', 8, 2, 128, 10, 30)
    # the 4-level P6 graph of the reference's own hub files (yolov5m6-multihead.yaml), at 'n' widths
    gen_stages('n6_128', 'n6', 3, 2, 128, full=False)
    gen_train('n6_128', 'n6', 3, 2, 128, 4, 12)
    gen_f3()
    gen_nms_options()
    gen_masks()
    gen_scale_img()
    gen_confusion()
    gen_train('n_64_ragged', 'n', 2, 2, 64, 3, 8, empty_first=True)
    gen_seg()
    gen_train('c1_640', 'n', 2, 4, 640, 50, 400)
    gen_eval_compact('c1_640', 'n', 2, 4, 640)
    gen_eval_compact('s_640', 's', 8, 2, 640)
    gen_train('m_128', 'm', 8, 2, 128, 10, 30)
    gen_eval_compact('m_128', 'm', 8, 2, 128)


if __name__ == '__main__':
    main()
