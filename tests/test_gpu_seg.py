"""SURVEY §8 row f4 on the MI355X: the segmentation-branch kernels (csrc/seg.hip) against plain torch, PanopticFeatureConnector /
PanopticSeg against the reference's own classes (tests/golden/seg.npz), and HNet's mixed detection + segmentation step against the
CPU oracle (oracle/ref_net.py + oracle/seg_ref.py)."""
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from hd_yolo_amd import ops, synth
from hnet.hnet import HNet
from hnet.segmentation import PanopticFeatureConnector, PanopticSeg
from oracle import seg_ref
from oracle.ref_net import RefNet

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda', 0)
G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'seg.npz'))
TOL = {torch.float32: 1e-4, torch.bfloat16: 1.5e-2}


def relmax(got, ref):
    return ((got - ref).abs().max() / (ref.abs().max() + 1e-12)).item()


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('shape,G_', [((2, 12, 10, 64), 32), ((3, 7, 5, 32), 32), ((1, 40, 24, 128), 32), ((2, 9, 9, 24), 4)])
def test_groupnorm_relu_forward_backward(shape, G_, dtype):
    N, H, W, C = shape
    x = (rnd(shape, 1, 2.0) + rnd((1, 1, 1, C), 2)).to(dtype)
    gamma, beta = rnd((C,), 3) + 1.2, rnd((C,), 4, 0.4)
    dout = rnd(shape, 5).to(dtype)
    xr = x.float().permute(0, 3, 1, 2).clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref = F.relu(F.group_norm(xr, G_, gr, br, 1e-5))
    ref.backward(dout.float().permute(0, 3, 1, 2))
    xd = torch.full((N, H, W, C + 16), 3.0, dtype=dtype, device=DEV)[..., 8:8 + C]       # a channel slice of a wider buffer
    xd.copy_(x)
    y, saved = ops.groupnorm_relu_fwd(xd, gamma.to(DEV), beta.to(DEV), G_)
    assert relmax(y.float().cpu(), ref.detach().permute(0, 2, 3, 1)) < TOL[dtype]
    dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dx = ops.groupnorm_relu_bwd(dout.to(DEV), xd, gamma.to(DEV), saved, G_, dg, db)
    assert relmax(dx.float().cpu(), xr.grad.permute(0, 2, 3, 1)) < TOL[dtype]
    assert relmax(dg.cpu(), gr.grad) < 2e-3 and relmax(db.cpu(), br.grad) < 2e-3          # fp32 sums of (bf16-rounded) products


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('inp,out', [((5, 7), (10, 14)), ((8, 12), (64, 96)), ((16, 16), (16, 16)), ((9, 4), (5, 11)), ((1, 6), (3, 6)),
                                     ((20, 40), (160, 320)), ((3, 300), (5, 2400))])      # x8 as the segmentation header resizes; a row beyond the LDS-staged W pass
def test_bilinear_align_corners_forward_backward(inp, out, dtype):
    N, C = 2, 16
    x = rnd((N, C) + inp, 7).to(dtype)
    xr = x.float().clone().requires_grad_(True)
    ref = F.interpolate(xr, size=out, mode='bilinear', align_corners=True)
    w = rnd(ref.shape, 8).to(dtype)
    ref.backward(w.float())
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    y = ops.bilinear_fwd(xd, out)
    assert relmax(y.float().cpu().permute(0, 3, 1, 2), ref.detach()) < TOL[dtype]
    base = rnd((N,) + out + (C,), 9).to(dtype).to(DEV)
    acc = ops.bilinear_fwd(xd, out, out=base.clone(), accumulate=True)
    assert relmax(acc.float().cpu(), (base.float().cpu() + ref.detach().permute(0, 2, 3, 1))) < TOL[dtype]
    dx = ops.bilinear_bwd(w.permute(0, 2, 3, 1).contiguous().to(DEV), inp)
    assert relmax(dx.float().cpu().permute(0, 3, 1, 2), xr.grad) < TOL[dtype]


@pytest.mark.parametrize('nc,weights,hw', [(3, [1.0, 2.0, 0.5], (37, 29)), (1, None, (37, 29)), (8, None, (37, 29)),
                                          (3, None, (173, 211)), (6, None, (160, 200)), (12, None, (90, 70))])     # the large ones: several pixels in flight per thread
def test_softmax_soft_dice_loss_and_gradient(nc, weights, hw):
    N, (H, W) = 3, hw
    kp = (nc + 7) // 8 * 8
    logits = rnd((N, H, W, kp), 11, 3.0)
    logits[..., nc:] = 0
    lab = torch.randint(0, max(nc, 2), (N, H, W), generator=torch.Generator().manual_seed(12))
    masks = F.one_hot(lab, max(nc, 2)).permute(0, 3, 1, 2).float()[:, :nc].contiguous()
    lr = logits[..., :nc].permute(0, 3, 1, 2).clone().requires_grad_(True)
    ref = 1 + seg_ref.soft_dice_criterion(torch.softmax(lr, 1), masks, weights)
    (ref * 1.7).backward()
    cw = None if weights is None else torch.tensor(weights, device=DEV)
    loss, dl = ops.softdice(logits.to(DEV), masks.to(DEV), cw, upstream=torch.tensor([1.7], device=DEV), want_grad=True)
    assert abs(loss.item() - ref.item()) < 1e-5
    assert relmax(dl.cpu()[..., :nc].permute(0, 3, 1, 2), lr.grad) < 1e-4 and (kp == nc or dl[..., nc:].abs().max().item() == 0)
    probs = ops.softmax2d(logits.to(DEV), nc)
    assert relmax(probs.cpu().permute(0, 3, 1, 2), torch.softmax(lr.detach(), 1)) < 1e-6


def _load(module, prefix):
    sd = {k[len(prefix):]: torch.from_numpy(G[k]) for k in G.files if k.startswith(prefix)}
    missing = module.load_state_dict(sd, strict=True)
    return module


def test_connector_matches_the_reference_class():
    names = OrderedDict((k, k) for k in ('23', '26', '29', '32'))
    con = _load(PanopticFeatureConnector([32, 64, 96, 128], 32, names), 'con_p_').to(DEV)
    feats = OrderedDict((n, torch.from_numpy(G[f'con_in_{n}']).to(DEV)) for n in names)
    y = con(feats)['0']
    assert relmax(y.float().cpu(), torch.from_numpy(G['con_out'])) < 1e-4


@pytest.mark.parametrize('dtype,tol', [(torch.float32, 2e-4), (torch.bfloat16, 8e-2)])
def test_panoptic_seg_matches_the_reference_class(dtype, tol):
    """loss, probabilities, feature gradients and parameter gradients of PanopticSeg against the reference's own PanopticSeg run
    (fp32: 1e-4-level; bf16 operands: the usual 2^-7-level agreement)."""
    cfg = {'in_channels': 64, 'num_classes': 3, 'feature_maps': OrderedDict((k, k) for k in ('17', '20', '23')), 'scale_factor': 8,
           'resize_mode': 'bilinear', 'class_weight': [1.0, 2.0, 0.5], 'roi_size': None}
    head = _load(PanopticSeg(cfg), 'seg_p_').to(DEV).train()
    feats = OrderedDict((n, torch.from_numpy(G[f'seg_in_{n}']).to(DEV).requires_grad_(True)) for n in cfg['feature_maps'])
    masks = torch.from_numpy(G['seg_masks']).to(DEV)
    H, W = masks.shape[-2:]
    targets = [[{'roi': torch.tensor([0.0, 0.0, W, H]), 'masks': masks[i]}] for i in range(2)]
    _, losses = head(feats, (H, W), None, targets, dtype=dtype)
    loss = losses['soft_iou_loss']
    assert abs(loss.item() - float(G['seg_loss'][0])) < (2e-5 if dtype == torch.float32 else 5e-3)
    loss.backward()
    # fp32: max error relative to the tensor's max.  bf16: relative L2 error — a ReLU whose bf16 pre-activation lands on the other side
    # of zero flips one element's gradient entirely: with ~0.3 % of the pre-activations within bf16 rounding of zero that alone is
    # sqrt(0.003) = 5 % of the gradient's L2 norm (measured: 0.7-5 % per tensor), and far more in a max-norm
    def err(got, ref):
        return relmax(got, ref) if dtype == torch.float32 else ((got - ref).norm() / ref.norm()).item()
    errs = {n: err(f.grad.float().cpu(), torch.from_numpy(G[f'seg_din_{n}'])) for n, f in feats.items()}
    errs.update({k: err(p.grad.cpu(), torch.from_numpy(G[f'seg_g_{k}'])) for k, p in head.named_parameters()})
    assert max(errs.values()) < tol, errs
    head.eval()
    with torch.no_grad():
        res, _ = head(OrderedDict((n, f.detach()) for n, f in feats.items()), (H, W), None, None, dtype=dtype)
    probs = torch.cat(res)
    assert tuple(probs.shape) == (2, 3, H, W)
    assert (probs.cpu() - torch.from_numpy(G['seg_eval_probs'])).abs().max().item() < (1e-4 if dtype == torch.float32 else 3e-2)
    with pytest.raises(Exception):
        head.train()(feats, (H, W), None, [[{'roi': torch.tensor([0.0, 0.0, W / 2, H]), 'masks': masks[i]}] for i in range(2)], dtype=dtype)


def _hnet(nc=2, ncls=3):
    cfg = {'backbone': {'type': 'yolov5', 'cfg': synth.make_cfg('n', nc), 'hyp': synth.make_hyp()},
           'headers': {'seg': {'type': 'PanopticSeg', 'configs': {'num_classes': ncls, 'feature_maps': None, 'in_channels': None, 'scale_factor': 8,
                                                                   'resize_mode': 'bilinear', 'class_weight': None, 'roi_size': None}}}}
    m = HNet(cfg)
    m.detector.load_state_dict(synth.synth_state_dict(synth.shapes_of(m.detector), seed=0), strict=False)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for k, p in m.headers.named_parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (0.05 if p.dim() == 4 else 0.3) + (1.0 if p.dim() == 1 and k.endswith('weight') else 0.0))
    return m


def test_hnet_mixed_detection_and_segmentation_step_matches_oracle():
    """One training step of HNet (metayolo backbone + pyramid + Detect header + PanopticSeg header on the pyramid taps): both losses and
    the gradients of backbone, neck, detection and segmentation parameters against the CPU oracle's autograd of the same graph."""
    B, S, nc, ncls = 2, 64, 2, 3
    m = _hnet(nc, ncls).to(DEV).train()
    x = synth.synth_images(B, S, seed=3)
    det_t = synth.synth_targets(B, S, nc, nmin=3, nmax=8, seed=4)
    lab = torch.randint(0, ncls, (B, S, S), generator=torch.Generator().manual_seed(6))
    masks = F.one_hot(lab, ncls).permute(0, 3, 1, 2).float().contiguous()
    targets = []
    for i, t in enumerate(det_t):
        anns = dict(t['anns'])
        anns['seg'] = [{'roi': torch.tensor([0.0, 0.0, S, S]), 'masks': masks[i].to(DEV)}]
        targets.append({**t, 'anns': anns})
    losses, _ = m(x.to(DEV), targets)
    assert set(losses) >= {'det_det_loss', 'seg_soft_iou_loss'}
    total = losses['det_det_loss'] + 2.0 * losses['seg_soft_iou_loss']
    total.backward()
    # oracle
    net = RefNet(synth.make_cfg('n', nc), synth.make_hyp())
    sd = net.init_state()
    for k, t in sd.items():
        if 'running' not in k:
            t.requires_grad_(True)
    seg_sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.headers['seg'].state_dict().items()}
    feats = net.features(sd, x, training=True)
    dets = net.det_logits(sd, {k: feats[k] for k in net.head['f']})
    det_loss, _ = net.det_loss(dets, synth.synth_targets(B, S, nc, nmin=3, nmax=8, seed=4))
    _, seg_loss = seg_ref.panoptic(seg_sd, OrderedDict((str(k), feats[k]) for k in (17, 20, 23)), 8, masks)
    (det_loss + 2.0 * seg_loss).backward()
    assert abs(losses['det_det_loss'].item() - det_loss.item()) < 2e-4 * abs(det_loss.item())
    assert abs(losses['seg_soft_iou_loss'].item() - seg_loss.item()) < 2e-5
    params = dict(m.detector.named_parameters())
    for k in ('backbone.0.conv.weight', 'backbone.4.cv1.conv.weight', 'neck.7.cv3.conv.weight', 'neck.13.m.0.cv2.conv.weight', 'neck.13.cv3.bn.weight',
              'headers.det.m.0.weight'):
        assert relmax(params[k].grad.cpu(), sd[k].grad) < 1e-3, k
    for k, p in m.headers['seg'].named_parameters():
        assert relmax(p.grad.cpu(), seg_sd[k].grad) < 1e-3, k
    # a step without the segmentation loss: its parameters' gradients are zero, the detector's are the plain detector's
    for p in m.parameters():
        p.grad = None
    losses, _ = m(x.to(DEV), targets)
    losses['det_det_loss'].backward()
    assert all(float(p.grad.abs().max()) == 0.0 for p in m.headers['seg'].parameters())


@pytest.mark.parametrize('half', [False, True])
def test_taped_segmentation_steps_equal_the_eager_ones(half):
    """hd_yolo_amd/segrun.py replays the segmentation branch's training forward / backward as recorded launch lists from the second step on
    (same feature-map addresses, parameters and gradient views).  Four optimizer steps with the tapes against four steps of the eager code on an
    identical copy: every loss and every parameter bit for bit — a replay that read a stale packed weight, skipped a launch or wrote a buffer
    the next step still needs would show within these steps.  Also: the tapes really are replayed (one record per step is not enough)."""
    from hd_yolo_amd import segrun
    from hd_yolo_amd.optim import SGD
    B, S, nc, ncls = 2, 64, 2, 3
    x = synth.synth_images(B, S, seed=3).to(DEV)
    det_t = synth.synth_targets(B, S, nc, nmin=3, nmax=8, seed=4)
    lab = torch.randint(0, ncls, (B, S, S), generator=torch.Generator().manual_seed(6))
    masks = F.one_hot(lab, ncls).permute(0, 3, 1, 2).float().contiguous().to(DEV)
    targets = []
    for i, t in enumerate(det_t):
        anns = {k: [{kk: (vv.to(DEV) if torch.is_tensor(vv) else vv) for kk, vv in a.items()} for a in v] for k, v in t['anns'].items()}
        anns['seg'] = [{'roi': torch.tensor([0.0, 0.0, S, S]), 'masks': masks[i]}]
        targets.append({**t, 'anns': anns})
    results, replays = {}, {}
    saved = segrun.TAPE
    try:
        for mode in (True, False):
            segrun.TAPE = mode
            m = _hnet(nc, ncls).to(DEV).train()
            if half:
                m.half()
            opt = SGD(m.parameters(), lr=1e-3, momentum=0.9, nesterov=True)
            hist = []
            for step in range(4):
                losses, _ = m(x, targets)
                (losses['det_det_loss'] + 2.0 * losses['seg_soft_iou_loss']).backward()
                opt.step()
                opt.zero_grad(set_to_none=True)
                hist.append((float(losses['det_det_loss']), float(losses['seg_soft_iou_loss'])))
            results[mode] = (hist, {k: p.detach().clone() for k, p in m.named_parameters()})
            run = next(iter(m.headers['seg'].__dict__['_hdy_runs'].values()))
            replays[mode] = (run.__dict__.get('_fwd') is not None and run._fwd['tape'].prog is not None,
                             run.__dict__.get('_bwd') is not None and run._bwd['tape'].prog is not None)
    finally:
        segrun.TAPE = saved
    assert replays[True] == (True, True) and replays[False] == (False, False), replays
    assert results[True][0] == results[False][0], (results[True][0], results[False][0])
    assert all(np.isfinite(v) for pair in results[True][0] for v in pair) and results[True][0][0] != results[True][0][3]        # the steps did train
    for k, p in results[True][1].items():
        assert torch.equal(p, results[False][1][k]), k


def test_segmentation_loss_is_the_callers_own_and_a_stale_backward_is_refused():
    """the header keeps one executor (activation buffers, loss buffers, tapes) per arithmetic type.  (a) the returned loss must not alias
    the executor's loss buffer — a history of `loss.detach()` kept across steps would all read the latest value; (b) a backward whose forward has been
    overtaken by another forward of the same header must raise, not silently use the later activations."""
    B, S, nc, ncls = 2, 64, 2, 3
    x1, x2 = synth.synth_images(B, S, seed=3).to(DEV), synth.synth_images(B, S, seed=9).to(DEV)
    det_t = synth.synth_targets(B, S, nc, nmin=3, nmax=8, seed=4)
    lab = torch.randint(0, ncls, (B, S, S), generator=torch.Generator().manual_seed(6))
    masks = F.one_hot(lab, ncls).permute(0, 3, 1, 2).float().contiguous().to(DEV)
    targets = []
    for i, t in enumerate(det_t):
        anns = {k: [{kk: (vv.to(DEV) if torch.is_tensor(vv) else vv) for kk, vv in a.items()} for a in v] for k, v in t['anns'].items()}
        anns['seg'] = [{'roi': torch.tensor([0.0, 0.0, S, S]), 'masks': masks[i]}]
        targets.append({**t, 'anns': anns})
    m = _hnet(nc, ncls).to(DEV).train()
    l1, _ = m(x1, targets)
    kept = l1['seg_soft_iou_loss'].detach()
    v1 = float(kept)
    l1['seg_soft_iou_loss'].backward()
    m.zero_grad(set_to_none=True)
    l2, _ = m(x2, targets)
    assert float(l2['seg_soft_iou_loss']) != v1
    assert float(kept) == v1                                  # the first step's loss still reads the first step's value
    # a second forward before the first one's backward: the stale backward is refused with a message that names the rule
    l3, _ = m(x1, targets)
    with pytest.raises(RuntimeError, match='overtaken'):
        l2['seg_soft_iou_loss'].backward()
    l3['seg_soft_iou_loss'].backward()                        # the latest forward's backward still runs


def test_hnet_eval_outputs():
    m = _hnet().to(DEV).eval()
    x = synth.synth_images(2, 64, seed=3).to(DEV)
    with torch.no_grad():
        losses, outputs = m(x)
    assert losses == {} and len(outputs['det']) == 2 and 'boxes' in outputs['det'][0]
    assert len(outputs['seg']) == 2 and tuple(outputs['seg'][0].shape) == (1, 3, 64, 64)
    assert abs(float(outputs['seg'][0].sum(1).mean()) - 1.0) < 1e-5


@pytest.mark.parametrize('nc,size,low_w', [(3, (48, 256), 32), (4, (33, 1000), 125), (1, (20, 64), 8)])
def test_fused_dice_gradient_and_w_pass_equals_the_two_launches(nc, size, low_w):
    """hdy_softdice_wgrad (loss + gradient reduced along W inside the loss kernel: no full-resolution gradient tensor) against hdy_softdice
    followed by the W pass of the resize backward: the loss and the reduced gradient must be identical bit for bit"""
    N, (H, W) = 2, size
    logits = rnd((N, H, W, 4), 21, 3.0)
    logits[..., nc:] = 0
    lab = torch.randint(0, max(nc, 2), (N, H, W), generator=torch.Generator().manual_seed(22))
    masks = F.one_hot(lab, max(nc, 2)).permute(0, 3, 1, 2).float()[:, :nc].contiguous()
    cw = None if nc != 3 else torch.tensor([1.0, 2.0, 0.5], device=DEV)
    ld, md = logits.to(DEV), masks.to(DEV)
    assert ops.softdice_wgrad_ok(ld, nc, low_w)
    loss_a, dl = ops.softdice(ld, md, cw, want_grad=True)
    want = torch.empty((N, H, low_w, 4), dtype=torch.float32, device=DEV)
    from hd_yolo_amd import _lib
    _lib.call('hdy_bilinear_bwd_axis', dl.data_ptr(), 4, want.data_ptr(), 4, N * H, low_w, W, 1, 4, 0, ops.dcode(torch.float32), ops.stream_ptr())
    loss_b, dw = ops.softdice_wgrad(ld, md, cw, low_w)
    assert torch.equal(loss_a, loss_b) and torch.equal(dw, want)
