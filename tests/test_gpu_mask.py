"""SURVEY §8 row f2 (mask branch) on the MI355X: roi_align forward/backward, the Mask-RCNN head's forward and backward through the HIP
conv kernels, against the oracle's restatement of the third-party pieces (oracle/mask_ref.py) and torch autograd on the CPU."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F
from collections import OrderedDict

from hd_yolo_amd import maskhead, ops, synth
from oracle import mask_ref

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda', 0)


def relmax(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def _rois(gen, R, B, size):
    c = torch.rand((R, 2), generator=gen) * size
    wh = torch.rand((R, 2), generator=gen) * size * 0.2 + 2
    r = torch.cat([torch.randint(0, B, (R, 1), generator=gen).float(), c - wh / 2, c + wh / 2], 1)
    r[0, 1:] = torch.tensor([-20.0, -15.0, 10.0, 12.0])         # partly outside the image
    r[1, 1:] = torch.tensor([size - 6.0, size - 5.0, size + 30.0, size + 40.0])
    r[2, 1:] = torch.tensor([40.0, 40.0, 40.2, 40.1])           # narrower than one feature pixel: clamped to 1
    return r


@pytest.mark.parametrize('dtype,tol', [(torch.float32, 1e-5), (torch.bfloat16, 1e-2)])
@pytest.mark.parametrize('aligned', [False, True])
def test_roi_align_forward_backward(dtype, tol, aligned):
    g = torch.Generator().manual_seed(3)
    B, C, H, W, stride, P = 3, 16, 20, 24, 8, 14
    feat = torch.randn((B, C, H, W), generator=g).to(dtype).float()
    rois = _rois(g, 37, B, H * stride)
    fr = feat.clone().requires_grad_(True)
    ref = mask_ref.roi_align(fr, rois, P, 1.0 / stride, 2, aligned)
    dout = torch.randn(ref.shape, generator=g).to(dtype).float()
    ref.backward(dout)
    # device: NHWC feature inside a wider (pitched) buffer
    buf = torch.zeros((B, H, W, C + 8), dtype=dtype, device=DEV)
    buf[..., :C] = feat.permute(0, 2, 3, 1).to(DEV).to(dtype)
    out = ops.roi_align(buf[..., :C], rois.to(DEV), 1.0 / stride, P, 2, aligned)
    assert relmax(out.permute(0, 3, 1, 2), ref.detach()) < tol
    df = ops.roi_align_bwd(dout.permute(0, 2, 3, 1).contiguous().to(DEV).to(dtype), (B, H, W, C), rois.to(DEV), 1.0 / stride, 2, aligned)
    assert relmax(df.permute(0, 3, 1, 2), fr.grad) < 1e-4        # fp32 atomics: order-dependent rounding only
    gbuf = torch.full((B, H, W, C + 8), 2.0, dtype=dtype, device=DEV)
    ops.cast_store(df, gbuf[..., :C], accumulate=True)
    assert relmax(gbuf[..., :C].permute(0, 3, 1, 2), fr.grad + 2.0) < max(tol, 1e-4) and (gbuf[..., C:] == 2.0).all()
    assert ops.roi_align(buf[..., :C], torch.zeros((0, 5), device=DEV), 0.125, P).shape == (0, P, P, C)


class _SegH(nn.Sequential):
    def __init__(self, c_in, nc_masks):
        super().__init__(OrderedDict([('maskrcnn_heads', mask_ref.MaskRCNNHeads(c_in, (256, 256, 256, 256), 1)),
                                      ('maskrcnn_preds', mask_ref.MaskRCNNPredictor(256, 256, nc_masks))]))


# bf16: every intermediate gradient of the 6-layer chain is rounded to bf16 and ReLU masks of near-zero activations may flip;
# fp32 is the parity mode
@pytest.mark.parametrize('dtype,tol', [(torch.float32, 2e-4), (torch.bfloat16, 8e-2)])
def test_mask_head_forward_backward(dtype, tol):
    torch.manual_seed(1)
    R, C, P, ncm = 9, 64, 14, 3
    seg_h = _SegH(C, ncm)
    for p in seg_h.parameters():
        p.data = p.data.to(dtype).float()
        if p.dim() == 1:
            p.data.uniform_(-0.2, 0.2)
    x = (torch.randn((R, C, P, P)) * 0.5).to(dtype).float()
    xr = x.clone().requires_grad_(True)
    ref = seg_h(xr)                                            # (R, ncm, 28, 28)
    dl = torch.randn(ref.shape) * 0.1
    ref.backward(dl)
    dev_h = _SegH(C, ncm).to(DEV)
    dev_h.load_state_dict(seg_h.state_dict())
    run = maskhead.MaskHeadRun(dev_h, dtype)
    out = run.forward(x.permute(0, 2, 3, 1).contiguous().to(DEV).to(dtype), train=True)
    assert out.dtype == torch.float32 and tuple(out.shape) == (R, 2 * P, 2 * P, ncm)
    assert relmax(out.permute(0, 3, 1, 2), ref.detach()) < tol
    grads = {id(p): torch.zeros_like(p) for p in dev_h.parameters()}
    dx = run.backward(dl.permute(0, 2, 3, 1).contiguous().to(DEV), lambda p: grads[id(p)])
    assert relmax(dx.permute(0, 3, 1, 2), xr.grad) < tol * 2
    for (name, p), q in zip(seg_h.named_parameters(), dev_h.parameters()):
        assert relmax(grads[id(q)], p.grad) < tol * 2, name


# ------------------------------------------------------------------------------------------ whole model against the reference
import os  # noqa: E402

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'masks.npz'), allow_pickle=False)


def _mask_model(hyp):
    from metayolo.models.yolo import Model
    cfg = synth.make_cfg('n', 2)
    cfg['headers'][0][3][3] = 1                         # masks: every class uses mask channel 1 (+ the general channel 0)
    m = Model(cfg, hyp)
    missing = m.load_state_dict(synth.mask_state_dict(m), strict=False)
    assert not missing.unexpected_keys
    return m.to(DEV)


def test_eval_masks_match_reference():
    """Detections + 28 x 28 masks of yolov5n / masks=1 on 2 x 128^2 tiles (tests/golden/masks.npz: the reference's own
    multiscale_roi_align / seg convs / score logic on the oracle's roi_align and Mask R-CNN head)."""
    model = _mask_model(synth.make_hyp(conf_thres=0.05)).eval()
    x = synth.synth_images(2, 128, seed=7).to(DEV)
    with torch.no_grad():
        _, outputs = model(x, compute_masks=True)
        plan = next(iter(model._eng().plans.values()))
        for l, f in enumerate(plan.mask_features()):
            assert relmax(f.permute(0, 3, 1, 2), torch.from_numpy(G[f'eval_maskmap_{l}'])) < 1e-4
    probs = torch.from_numpy(G['eval_mask_probs'])
    off = 0
    for b, o in enumerate(outputs):
        o = o['det']
        n = len(G[f'eval_{b}_boxes'])
        np.testing.assert_allclose(o['boxes'].cpu().numpy(), G[f'eval_{b}_boxes'], rtol=1e-4, atol=1e-3)
        assert np.array_equal(o['labels'].cpu().numpy(), G[f'eval_{b}_labels'])
        labels = torch.from_numpy(G[f'eval_{b}_labels']).clamp(min=0)
        want = probs[off:off + n][torch.arange(n), model.headers['det'].mask_indices.cpu()[labels]][:, None]
        assert o['masks'].shape == (n, 1, 28, 28)
        assert (o['masks'].cpu() - want).abs().max() < 2e-4
        off += n
    _, plain = model(x, compute_masks=False)
    assert 'masks' not in plain[0]['det']


@pytest.mark.parametrize('fused', ['1', '0'])
def test_train_step_with_mask_loss_matches_reference(fused, monkeypatch):
    """det + mask loss and every parameter gradient (3 numbers each) of one training step with anchor-shaped truths; with the
    fused detection-loss kernel (the mask branch then re-derives the matched cells with the tensor-expression matcher) and without."""
    monkeypatch.setenv('HDY_FUSED_LOSS', fused)
    model = _mask_model(synth.make_hyp()).train()
    x = synth.synth_images(2, 128, seed=11).to(DEV)
    targets = synth.synth_mask_targets(2, 128, 2, per_image=6, seed=4)
    losses, _ = model(x, targets, compute_masks=True)
    l = losses['det']
    np.testing.assert_allclose(l['det_loss'].detach().cpu().numpy(), G['train_det_loss'], rtol=2e-4)
    np.testing.assert_allclose(l['mask_loss'].detach().cpu().numpy(), G['train_mask_loss'], rtol=2e-4)
    (l['det_loss'] + l['mask_loss']).backward()
    params = dict(model.named_parameters())
    worst = 0.0
    for name, ref in zip(G['gradsum_names'].tolist(), G['gradsum']):
        g = params[name].grad.double()
        got = np.array([g.sum().item(), g.abs().sum().item(), g.pow(2).sum().sqrt().item()])
        worst = max(worst, abs(got[2] - ref[2]) / (ref[2] + 1e-12))
        # 5e-3: the mask path adds fp32-atomic scatter order and BatchNorm over as few as 32 samples per channel (2 tiles, 4x4 P5)
        assert abs(got[2] - ref[2]) <= 5e-3 * ref[2] + 1e-7, (name, got, ref)
        assert abs(got[1] - ref[1]) <= 5e-3 * ref[1] + 1e-6, (name, got, ref)
    for k in [k for k in G.files if k.startswith('grad:')]:
        assert relmax(params[k[5:]].grad, torch.from_numpy(G[k])) < 5e-3, k
    # a step without mask loss leaves the mask branch's gradients at zero instead of stale
    model.zero_grad(set_to_none=True)
    losses, _ = model(x, targets, compute_masks=False)
    losses['det']['det_loss'].backward()
    assert float(params['headers.det.seg_h.maskrcnn_heads.mask_fcn1.weight'].grad.abs().sum()) == 0.0
    assert float(params['headers.det.seg.0.conv.weight'].grad.abs().sum()) == 0.0


@pytest.mark.parametrize('min_iou', [0.8, 0.3, 0.0])
def test_mask_selection_on_the_device_equals_the_tensor_expressions(min_iou):
    """hdy_mask_select (same candidates as the loss kernel's matcher, decode kernel's boxes, IoU in input pixels, best cell per target in the
    reference's row order, compaction per level) against the tensor-expression selection of Detect.mask_losses (reference:
    yolo_head.py:231-262) on the same logits: kept targets, per-level rois and the order permutation must be identical, bit for bit."""
    from hd_yolo_amd import engine as _engine
    from metayolo.models.utils_general import paired_box_iou, xywh2xyxy
    model = _mask_model(synth.make_hyp()).train()
    head = model.headers['det']
    x = synth.synth_images(3, 256, seed=5).to(DEV)
    targets = [t['anns']['det'][0] for t in synth.synth_mask_targets(3, 256, 2, per_image=25, seed=6)]
    gts, gt_labels = head.flatten_targets(targets, DEV)
    gts, tcls = gts.contiguous(), gt_labels[:, 1:].float().contiguous()
    eng = model._eng()
    plan, _, _ = eng.forward_fused_loss(x, _engine.compute_dtype(model, x), head, gts, tcls)
    apx = [v for i in range(head.nl) for v in head._anchor_px_cached(i)]
    counts, keep_t, rois, order = plan.fused_loss(head).mask_select(gts, apx, [head._stride_cached(i) for i in range(head.nl)], min_iou)
    counts = counts.tolist()
    # the tensor-expression selection
    dets = [d.detach() for d in plan.det_views()]
    _, tids, indices, _ = head.matcher(dets, gts)
    preds = head.compute_proposals(dets)
    props, gt_props, obj_ids, all_rois = [], [], [], []
    for i, buf in enumerate(head.anchors):
        v = plan.mask_vals[i]
        b, a, gj, gi = indices[i]
        props.append(xywh2xyxy(preds[i][b, a, gj, gi, :4]))
        gb = xywh2xyxy(gts[tids[i]][:, 1:] * gts.new([v.w, v.h, v.w, v.h])) * buf.stride
        gt_props.append(gb)
        obj_ids.append(tids[i])
        all_rois.append(torch.cat([b[:, None].to(gb.dtype), gb], -1))
    sizes = [len(o) for o in obj_ids]
    props, gt_props, obj_ids, all_rois = torch.cat(props), torch.cat(gt_props), torch.cat(obj_ids), torch.cat(all_rois)
    ious = paired_box_iou(props, gt_props)
    nobj = int(gts.shape[0])
    best = torch.zeros(nobj, dtype=ious.dtype, device=DEV).scatter_reduce_(0, obj_ids, ious, 'amax', include_self=False)
    rows = torch.arange(len(ious), device=DEV)
    hit = ious == best[obj_ids]
    arg = torch.full((nobj,), len(ious), dtype=torch.long, device=DEV).scatter_reduce_(0, obj_ids[hit], rows[hit], 'amin')
    present = torch.zeros(nobj, dtype=torch.bool, device=DEV).index_fill_(0, obj_ids, True)
    keep = arg[(best >= min_iou) & present]
    level = torch.repeat_interleave(torch.arange(head.nl, device=DEV), torch.tensor(sizes, device=DEV))
    klev = level[keep]
    assert counts[0] == len(keep) and (min_iou > 0.5 or len(keep) > 20)
    assert torch.equal(keep_t[:counts[0]], obj_ids[keep])
    for l in range(head.nl):
        want = all_rois[keep[klev == l]]
        assert counts[1 + l] == len(want) and torch.equal(rois[l, :len(want)], want), l
    pos = torch.cat([(klev == l).nonzero().flatten() for l in range(head.nl)])
    want_order = torch.empty_like(pos)
    want_order[pos] = torch.arange(len(pos), device=DEV)
    assert torch.equal(order[:counts[0]], want_order)
