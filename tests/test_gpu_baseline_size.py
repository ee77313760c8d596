"""Whole-model runs of BASELINE.json configs[2..4] at their per-GPU size: the oracle cannot run these sizes in seconds, so the
checks are the size-independent properties the domain offers — finite loss and a gradient for every parameter, the kernel families the plan is
meant to pick (dispatch log), determinism across two runs, NMS idempotence and descending objectness, and per-tile independence of the
inference path (the last two tiles of the batch against the same two tiles run alone).

Reference callers: train.py:457-478 (C3 train step), val_nuclei.py:127-144 (C4 inference loop), hnet/hnet.py:104-265 (C5 forward)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from hd_yolo_amd import _lib, ops, synth  # noqa: E402

DEV = 'cuda:0'


def _iou(a, b):
    lt = torch.maximum(a[:, None, :2], b[None, :, :2])
    rb = torch.minimum(a[:, None, 2:], b[None, :, 2:])
    inter = (rb - lt).clamp(min=0).prod(2)
    area = lambda t: (t[:, 2] - t[:, 0]) * (t[:, 3] - t[:, 1])
    return inter / (area(a)[:, None] + area(b)[None, :] - inter + 1e-9)


def _train_step(model, x, targets):
    for p in model.parameters():
        p.grad = None
    _lib.dispatch_log(reset=True)
    losses, _ = model(x, targets)
    loss = losses['det']['det_loss']
    loss.backward()
    torch.cuda.synchronize()
    return float(loss.detach()), {k: p.grad.detach().clone() for k, p in model.named_parameters()}, set(_lib.dispatch_log(reset=True))


@pytest.mark.timeout(900)
def test_c3_yolov5m_batch32_640_train_step():
    """BASELINE configs[2] per GPU: yolov5m, 8 classes, 32 tiles of 640x640, bf16 operands.  Channel widths 48 / 96 / 192 / 384 / 768: rows that are not
    multiples of 128 bytes on the narrow layers, the deep-pipelined kernels on the 192- / 384- / 768-wide ones."""
    from metayolo.models.yolo import Model
    model = Model(synth.make_cfg('m', 8), synth.make_hyp())
    sd0 = synth.synth_state_dict(synth.shapes_of(model), seed=0)
    model.load_state_dict(sd0, strict=False)
    model = model.to(DEV).train()
    model.half()
    x = synth.synth_images(32, 640, seed=0).to(DEV)
    targets = synth.synth_targets(32, 640, 8, seed=1)
    rm0 = {k: v.clone() for k, v in model.state_dict().items() if k.endswith('running_mean')}
    loss1, g1, log1 = _train_step(model, x, targets)
    assert np.isfinite(loss1) and loss1 > 0
    bad = [k for k, g in g1.items() if g is None or not torch.isfinite(g).all() or float(g.abs().max()) == 0.0]
    assert not bad, f'parameters without a usable gradient: {bad[:5]}'
    # every BatchNorm saw the batch
    assert all(not torch.equal(v, model.state_dict()[k]) for k, v in rm0.items())
    # kernel families a yolov5m step at this size is meant to run on
    want = {'conv_stem', 'deep_256x128', 'wgrad_deep', 'wgrad3x3', 'conv1x1_bwd_96'}     # (conv1x1_bwd_96: the fused 1x1 backward's 96-channel instance, round 6)           # (the 48-channel stem is outside the fused stem weight gradient's 16 / 32 / 64)
    assert any(n.startswith('wgrad_stem') for n in log1) and want <= log1, f'kernel families missing from the C3 step: {sorted(want - log1)}; ran {sorted(log1)}'
    plan = next(iter(model._eng().plans.values()))
    assert all(torch.isfinite(d).all() for d in plan.det_views())
    # determinism: the same step again from the same weights and running statistics (fixed-order slab reductions, no floating-point atomics on the
    # gradient path) gives the same gradient bits; the loss SUM goes through fp32 atomics over workgroup partials and may move in the last place
    model.load_state_dict(sd0, strict=False)
    loss2, g2, log2 = _train_step(model, x, targets)
    assert abs(loss1 - loss2) <= 1e-5 * abs(loss1), (loss1, loss2)
    diff = [k for k in g1 if not torch.equal(g1[k], g2[k])]
    assert not diff, f'gradients differ between two identical steps: {diff[:5]}'
    assert log1 == log2


@pytest.mark.timeout(1200)
def test_c4_yolov5l_batch128_1024_inference():
    """BASELINE configs[3]: yolov5l, 128 tiles of 1024x1024, bf16 network + decode + NMS + outputs (64 512 candidates per tile, 51 GB of activations)."""
    from metayolo.models.yolo import Model
    model = Model(synth.make_cfg('l', 8), synth.make_hyp())
    model.load_state_dict(synth.synth_state_dict(synth.shapes_of(model), seed=0), strict=False)
    model = model.to(DEV).eval().half()
    B, S = 128, 1024
    x = synth.synth_images(B, S, seed=0).to(DEV)
    head = model.headers['det']
    with torch.no_grad():
        # random-init yolov5l: logits of +-1e5, every sigmoid exactly 0 or 1 -> unit logit spread, then the threshold where tile 0 keeps ~1024 candidates
        # (dense nuclei: the filter, the sort and the greedy pass all have work)
        spread = synth.calibrate_det_logits(model, x[:2].contiguous())
        assert all(np.isfinite(s) and s > 0 for s in spread), spread
        model(x[:2].contiguous())
        plan2 = [pl for pl in model._eng().plans.values() if pl.det_views()[0].shape[0] == 2][-1]
        p2 = head.decode_all(plan2.det_views())
        obj = p2[0, :, 4].float()
        assert 0.0 < float(obj.min()) and float(obj.max()) < 1.0 and float(obj.std()) > 0.01
        head.nms_params = dict(head.nms_params, conf_thres=synth.dense_conf_thres(p2[0], 1024))
        _lib.dispatch_log(reset=True)
        _, outs = model(x)
        torch.cuda.synchronize()
        log = set(_lib.dispatch_log(reset=True))
        assert {'deep_256x256', 'deep_256x128', 'conv_stem', 'conv3x3_c64', 'conv3x3_c128'} <= log, sorted(log)      # (conv3x3_c128: the nine 128 -> 128 3x3 layers at 128 x 128, round 6)
        assert len(outs) == B
        n = [len(o['det']['boxes']) for o in outs]
        assert min(n) > 0 and max(n) <= int(head.nms_params['max_det']), (min(n), max(n))
        for o in outs:
            b = o['det']['boxes']
            assert torch.isfinite(b).all() and torch.isfinite(o['det']['scores']).all()
            assert (b[:, 2] - b[:, 0] >= 2).all() and (b[:, 3] - b[:, 1] >= 2).all()            # remove_small_boxes(min_size=2)
        # the kept rows come out in descending objectness, and running the NMS again on its own output keeps every row (idempotence)
        plan = [pl for pl in model._eng().plans.values() if pl.det_views()[0].shape[0] == B][-1]
        preds = head.decode_all(plan.det_views())
        assert preds.shape == (B, 64512, head.no + 1)
        p = head.nms_params
        res = ops.nms_batched(preds, head.nc, p['conf_thres'], p['iou_thres'], int(p['max_det']))
        nk = res['n_keep'].tolist()
        assert nk == n
        for b in (0, 1, B // 2, B - 2, B - 1):
            obj = res['scores'][b, :nk[b], 0]
            assert (obj[:-1] >= obj[1:]).all(), f'tile {b}: kept rows not in descending objectness'
            assert (obj > p['conf_thres']).all()
            rows = preds[b].index_select(0, res['keep'][b, :nk[b]])
            again = ops.nms_batched(rows[None].contiguous(), head.nc, p['conf_thres'], p['iou_thres'], int(p['max_det']))
            assert int(again['n_keep'][0]) == nk[b] and torch.equal(again['keep'][0, :nk[b]].cpu(), torch.arange(nk[b]))
        # determinism: the same batch again, bit for bit
        _, outs2 = model(x)
        for o, q in zip(outs, outs2):
            assert torch.equal(o['det']['boxes'], q['det']['boxes']) and torch.equal(o['det']['labels'], q['det']['labels'])
        # per-tile independence: the last two tiles alone (eval mode: BatchNorm folded, no batch statistics).  The small batch takes other kernel families for
        # some layers (fewer tiles), i.e. another bf16 rounding order: compare detections by overlap, not bit for bit.
        _, alone = model(x[B - 2:].contiguous())
        for o, q in zip(outs[B - 2:], alone):
            a, b = o['det']['boxes'].float(), q['det']['boxes'].float()
            iou = _iou(a, b)
            hit = (iou.max(1).values > 0.9).float().mean().item()
            assert hit > 0.9 and abs(len(a) - len(b)) <= 0.1 * len(a) + 2, (hit, len(a), len(b))
    del model, x, outs, preds
    torch.cuda.empty_cache()


@pytest.mark.timeout(900)
def test_c5_hnet_batch16_1280_train_step():
    """BASELINE configs[4] per GPU: hnet (yolov5s backbone + pyramid shared by Detect and PanopticSeg), 16 tiles of 1280x1280, mixed det + seg loss."""
    from hnet.hnet import HNet
    B, S, ncls = 16, 1280, 3
    cfg = {'backbone': {'type': 'yolov5', 'cfg': synth.make_cfg('s', 8), 'hyp': synth.make_hyp()},
           'headers': {'seg': {'type': 'PanopticSeg', 'configs': {'num_classes': ncls, 'feature_maps': None, 'in_channels': None, 'scale_factor': 8,
                                                                   'resize_mode': 'bilinear', 'class_weight': None, 'roi_size': None}}}}
    m = HNet(cfg)
    m.detector.load_state_dict(synth.synth_state_dict(synth.shapes_of(m.detector), seed=0), strict=False)
    m = m.to(DEV).train().half()
    x = synth.synth_images(B, S, seed=0).to(DEV)
    det_t = synth.synth_targets(B, S, 8, seed=1)
    g = torch.Generator().manual_seed(3)
    lab = torch.randint(0, ncls, (B, S // 16, S // 16), generator=g).repeat_interleave(16, 1).repeat_interleave(16, 2).to(DEV)
    masks = torch.nn.functional.one_hot(lab, ncls).permute(0, 3, 1, 2).float().contiguous()
    targets = []
    for i, t in enumerate(det_t):
        anns = {k: [{kk: (vv.to(DEV) if torch.is_tensor(vv) else vv) for kk, vv in a.items()} for a in v] for k, v in t['anns'].items()}
        anns['seg'] = [{'roi': torch.tensor([0.0, 0.0, S, S]), 'masks': masks[i]}]
        targets.append({**t, 'anns': anns})

    def step():
        for p in m.parameters():
            p.grad = None
        losses, _ = m(x, targets)
        (losses['det_det_loss'] + losses['seg_soft_iou_loss']).backward()
        torch.cuda.synchronize()
        return float(losses['det_det_loss'].detach()), float(losses['seg_soft_iou_loss'].detach()), {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}

    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    d1, s1, g1 = step()
    assert np.isfinite(d1) and np.isfinite(s1) and d1 > 0 and 0 < s1 < 1.0 + 1e-3        # soft dice loss of a 3-class problem lies in (0, 1]
    names = [k for k, p in m.named_parameters() if p.requires_grad]
    missing = [k for k in names if k not in g1 or not torch.isfinite(g1[k]).all()]
    assert not missing, f'parameters without a finite gradient: {missing[:5]}'
    assert any(k.startswith('headers.seg') or 'seg' in k for k in g1) and float(sum(g.abs().sum() for k, g in g1.items() if 'seg' in k)) > 0
    # the shared backbone receives both losses: its gradient differs from the detector-only gradient
    m.load_state_dict(sd0)
    d2, s2, g2 = step()
    assert abs(d1 - d2) <= 1e-5 * abs(d1) and abs(s1 - s2) <= 1e-5 * abs(s1), (d1, d2, s1, s2)
    k0 = 'detector.backbone.1.conv.weight' if 'detector.backbone.1.conv.weight' in g1 else next(k for k in g1 if 'backbone.1.conv.weight' in k)
    rel = float((g1[k0] - g2[k0]).abs().max() / (g1[k0].abs().max() + 1e-30))
    assert rel < 1e-3, f'{k0}: two identical hnet steps differ by {rel}'          # roi-free path; the resize backward is a gather (no atomics)
    del m, x, masks
    torch.cuda.empty_cache()
