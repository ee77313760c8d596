"""Whole-model parity on a real MI355X: metayolo.models.yolo.Model (HIP plans) against the reference's golden vectors
and against the CPU oracle (oracle/ref_net.py) on the same seeded weights, tiles and targets.

Tolerances (north star: fp32 boxes/logits within 1e-4 relative, NMS order bit-exact):
  fp32 mode  activations / logits 1e-4 of the tensor's max magnitude; decoded boxes rtol 1e-4 + 1e-3 px;
             loss rtol 2e-4; gradients 1e-3 of the per-tensor max (summation order differs: split-K slabs, MFMA chains);
  bf16 mode  logits 6e-2 of max after 24 layers of bf16 operands (stated, not a parity claim), loss within 3 %.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from hd_yolo_amd import synth  # noqa: E402

DEV = 'cuda:0'


def build(variant, nc, hyp=None):
    from metayolo.models.yolo import Model
    model = Model(synth.make_cfg(variant, nc), hyp or synth.make_hyp())
    sd = synth.synth_state_dict(synth.shapes_of(model), seed=0)
    missing = model.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys
    return model.to(DEV)


def ragged(targets, tag):
    """The *_ragged goldens were made with an empty first tile (no boxes, no labels)."""
    if tag.endswith('_ragged'):
        a = targets[0]['anns']['det'][0]
        a['boxes'], a['labels'] = a['boxes'][:0], a['labels'][:0]
    return targets


def relmax(got, ref):
    got, ref = torch.as_tensor(got).float().cpu(), torch.as_tensor(ref).float().cpu()
    return ((got - ref).abs().max() / (ref.abs().max() + 1e-12)).item()


def elementwise(got, ref, rtol=1e-4, arms=2e-5):
    """north_star's "within 1e-4 relative", element by element: the worst |error| - rtol * |ref| in units of rms(ref); <= arms passes.  The
    absolute term covers elements that cancel to near zero (a different summation order costs them ~1e-6 of the tensor's rms: the largest value on
    any tensor of these goldens is 1.0e-5, scripts/probes/elementwise_parity.py); `relmax` alone leaves small elements unconstrained."""
    got, ref = torch.as_tensor(got).float().cpu(), torch.as_tensor(ref).float().cpu()
    rms = ref.pow(2).mean().sqrt().item() + 1e-30
    return ((got - ref).abs() - rtol * ref.abs()).max().item() / rms


@pytest.mark.parametrize('tag,variant', [('n_64', 'n'), ('s_128', 's'), ('n6_128', 'n6')])
def test_eval_matches_reference_golden_fp32(golden_dir, tag, variant):
    g = np.load(os.path.join(golden_dir, f'stages_{tag}.npz'))
    batch, size, nc = (int(v) for v in g['meta'])
    model = build(variant, nc, synth.make_hyp(conf_thres=float(g['conf_thres']))).eval()
    x = synth.synth_images(batch, size, seed=7).to(DEV)
    with torch.no_grad():
        losses, outputs = model(x)
        eng = model._eng()
        plan = next(iter(eng.plans.values()))
        worst = 0.0
        worst_el = -1.0
        for k in g.files:
            if k.startswith('stage_'):
                worst = max(worst, relmax(plan.feature(int(k[6:])), g[k]))
                worst_el = max(worst_el, elementwise(plan.feature(int(k[6:])), g[k]))
            elif k.startswith('neck_'):
                worst = max(worst, relmax(plan.feature(int(k[5:])), g[k]))
                worst_el = max(worst_el, elementwise(plan.feature(int(k[5:])), g[k]))
        assert worst < 1e-4, f'feature maps: {worst:.2e}'
        assert worst_el <= 2e-5, f'feature maps, element by element: {worst_el:.2e} rms beyond 1e-4 * |ref|'
        dets = plan.det_views()
        assert len(dets) == sum(k.startswith('det_') for k in g.files)
        for i in range(len(dets)):
            assert relmax(dets[i], g[f'det_{i}']) < 1e-4
            assert elementwise(dets[i], g[f'det_{i}']) <= 2e-5, f'logits of level {i}, element by element'
        head = model.headers['det']
        preds = head.compute_proposals(dets)
        for i in range(len(dets)):
            np.testing.assert_allclose(preds[i].cpu().numpy(), g[f'pred_{i}'], rtol=1e-4, atol=1e-3)
    assert losses == {'det': {}}
    total = 0
    for b in range(batch):
        o = outputs[b]['det']
        rb, rs, rl = g[f'out_{b}_boxes'], g[f'out_{b}_scores'], g[f'out_{b}_labels']
        assert o['boxes'].shape == rb.shape, 'number of detections differs from the reference'
        np.testing.assert_allclose(o['boxes'].cpu().numpy(), rb, rtol=1e-4, atol=1e-3)
        np.testing.assert_allclose(o['scores'].cpu().numpy(), rs, rtol=1e-4, atol=1e-6)
        assert np.array_equal(o['labels'].cpu().numpy(), rl)
        total += len(rb)
    assert total > 0


@pytest.mark.parametrize('tag,variant', [('c1_640', 'n'), ('s_640', 's'), ('m_128', 'm')])
def test_full_size_eval_matches_reference_golden_fp32(golden_dir, tag, variant):
    """BASELINE config C1 exactly (yolov5n, 2 classes, batch 4, 640x640), yolov5s at 640x640 and yolov5m (config C3's widths: 48 / 96 /
    192 / 384 / 768 channels) at 128x128: detection logits (checksums + a strip) and the final per-image detections — 300 kept boxes each
    at 640x640, in the reference's NMS order — against the reference's own run."""
    g = np.load(os.path.join(golden_dir, f'eval_{tag}.npz'))
    batch, size, nc = (int(v) for v in g['meta'])
    model = build(variant, nc, synth.make_hyp(conf_thres=float(g['conf_thres']))).eval()
    x = synth.synth_images(batch, size, seed=7).to(DEV)
    with torch.no_grad():
        _, outputs = model(x)
        dets = next(iter(model._eng().plans.values())).det_views()
    for i, d in enumerate(dets):
        assert relmax(d[:1, :1, :4], g[f'det_{i}_strip']) < 1e-4
        d64 = d.double()
        got = np.array([d64.sum().item(), d64.abs().sum().item(), d64.pow(2).sum().sqrt().item()])
        np.testing.assert_allclose(got[1:], g[f'det_{i}_sums'][1:], rtol=1e-5)
        assert abs(got[0] - g[f'det_{i}_sums'][0]) < 1e-5 * g[f'det_{i}_sums'][1]
    for b in range(batch):
        o = outputs[b]['det']
        rb, rs, rl = g[f'out_{b}_boxes'], g[f'out_{b}_scores'], g[f'out_{b}_labels']
        n = len(rb)
        assert o['boxes'].shape == rb.shape and n == (300 if size == 640 else n) and n >= 40
        # 25 200 candidates per tile computed on different hardware: a pair of scores closer than the 1e-6 agreement of the logits may
        # swap places, an IoU within 1e-6 of the threshold may flip one decision.  Kept-order exactness is what the oracle tests pin (same
        # inputs bit for bit); here: the same boxes up to two per image, and identical rows wherever the order did not move
        gb = o['boxes'].cpu().numpy()
        d = np.abs(gb[:, None, :] - rb[None, :, :]).max(-1)
        assert (d.min(1) < 2e-3).sum() >= n - 2 and (d.min(0) < 2e-3).sum() >= n - 2, ((d.min(1) < 2e-3).sum(), (d.min(0) < 2e-3).sum())
        same = np.abs(gb - rb).max(-1) < 2e-3
        assert same.sum() >= n - 10
        np.testing.assert_allclose(o['scores'].cpu().numpy()[same], rs[same], rtol=1e-4, atol=1e-6)
        assert np.array_equal(o['labels'].cpu().numpy()[same], rl[same])


def _iou_matrix(a, b):
    x1, y1 = np.maximum(a[:, None, 0], b[None, :, 0]), np.maximum(a[:, None, 1], b[None, :, 1])
    x2, y2 = np.minimum(a[:, None, 2], b[None, :, 2]), np.minimum(a[:, None, 3], b[None, :, 3])
    inter = np.clip(x2 - x1, 0, None) * np.clip(y2 - y1, 0, None)
    aa, ab = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]), (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    return inter / (aa[:, None] + ab[None, :] - inter + 1e-12)


# fraction of the reference's kept detections (fp32 CPU run of the reference's own code) that the bf16 network reproduces: a detection of the same
# label at IoU >= 0.9.  Measured on MI355X first, then pinned a few points below (see the test's docstring); the floor is the assertion.
# Measured (MI355X, round 5): yolov5n / C1 0.996 per tile (0.993-0.997), mean best IoU 0.990; yolov5s 0.367 (0.35 / 0.383), mean best IoU 0.79 — with
# synth_state_dict's gain the 8-class yolov5s' objectness logits are nearly flat over a tile's 25 200 candidates, so WHICH 300 survive the
# threshold-and-NMS is decided by logit differences below bf16's resolution: the same sensitivity test_train_step_bf16_is_close_to_fp32 documents
# for its gradients.  The floors sit a few points under the measurements.
BF16_DETECTION_RECALL = {'c1_640': 0.97, 's_640': 0.30}      # 's_640': a sensitivity note, not a pinned result — see BF16_CALIBRATED_RECALL below


@pytest.mark.parametrize('tag,variant', [('c1_640', 'n'), ('s_640', 's')])
def test_full_size_eval_bf16_detections_against_reference_golden(golden_dir, tag, variant):
    """`model.half().eval()` — what val_nuclei.py:116,143 runs and what BASELINE configs[3] and bench.py's `infer` time — against the reference's
    fp32 detections at 640x640 (300 kept boxes per tile).  The fp32 goldens above exercise conv_igemm<float>; this run goes through the bf16 kernels
    the benchmark runs (asserted from the dispatch log: the deep-pipelined and the filter-resident 3x3 kernels for yolov5s).  bf16 operands move
    a logit by ~1e-2 of its scale: scores near the confidence threshold and near-ties in the NMS order change, so the comparison is a recall at
    IoU >= 0.9 with equal labels, its measured value printed and a floor asserted (the random-init network's objectness is nearly flat,
    which makes this the hard case: a trained detector separates its scores far more than a rounding does)."""
    from hd_yolo_amd import _lib
    g = np.load(os.path.join(golden_dir, f'eval_{tag}.npz'))
    batch, size, nc = (int(v) for v in g['meta'])
    model = build(variant, nc, synth.make_hyp(conf_thres=float(g['conf_thres']))).eval().half()
    x = synth.synth_images(batch, size, seed=7).to(DEV)
    # a 2-tile batch has 50 row tiles where the bench's 64 tiles have 1 600: HDY_DEEP_MIN_TILES = 1 gives the wide layers to the deep-pipelined kernel
    # the benchmark runs them on (set before the plan is built: the switch also steers its sizing queries)
    with _lib.option('HDY_DEEP_MIN_TILES', 1):
        _lib.dispatch_log(reset=True)
        with torch.no_grad():
            _, outputs = model(x)
        log = set(_lib.dispatch_log(reset=True))
    if variant == 's':
        assert any(n.startswith('deep_256x') for n in log) and 'conv3x3_c64' in log, sorted(log)
    fracs, half, ious = [], [], []
    for b in range(batch):
        o = outputs[b]['det']
        rb, rl = g[f'out_{b}_boxes'], g[f'out_{b}_labels']
        gb, gl = o['boxes'].float().cpu().numpy(), o['labels'].cpu().numpy()
        assert len(gb) > 0 and np.isfinite(gb).all()
        iou = _iou_matrix(rb, gb)
        iou[rl[:, None] != gl[None, :]] = 0.0
        best = iou.max(1)
        fracs.append(float((best >= 0.9).mean()))
        half.append(float((best >= 0.5).mean()))
        ious.append(float(best.mean()))
    frac = float(np.mean(fracs))
    print(f'bf16 detections vs reference fp32 goldens [{tag}]: recall@IoU0.9 per tile {[round(f, 3) for f in fracs]} mean {frac:.3f}, recall@IoU0.5 {np.mean(half):.3f}, '
          f'mean best IoU {np.mean(ious):.3f}, kept {[len(outputs[b]["det"]["boxes"]) for b in range(batch)]}')
    assert frac >= BF16_DETECTION_RECALL[tag], (frac, BF16_DETECTION_RECALL[tag])


# Round 6: fixtures whose scores SEPARATE.  tests/golden/make_golden.py `calibrated` rescales the reference model's three detection
# convs on the CPU so that every level's logits have unit spread (what a trained detector's look like; synth.calibrate_det_logits does the same for
# the C4 bench) and stores the factors; the test applies the same factors, so both sides run identical weights.  The flat-score 's_640' row above stays
# as a SENSITIVITY NOTE (0.30 floor: "mostly different boxes" there is decided below bf16 resolution and says nothing about the kernels); what
# protects the bf16 kernels is the unit-by-unit test further down, what protects bf16 DETECTIONS is this one.  Floors: measured, then pinned a few points under.
# Measured on MI355X (round 6): yolov5s @640 — logits 0.5-0.9 % relative L2 from the reference's fp32 logits, recall@IoU0.9 0.600, recall@IoU0.5 0.957, the reference's
# 100 best detections 0.965 at IoU 0.5; yolov5l @256 — logits 0.5-0.6 %, 0.917 / 0.923 / 0.970.  The LOGIT bound is the one that fails for the right reason (a wrong
# bf16 kernel moves it by orders of magnitude); the detection recalls say what that 1 % does downstream: with random weights neighbouring cells / anchors predict
# near-equal boxes and objectness, so NMS picks another representative of the same object (IoU 0.5-0.9) — recall@0.5 is 0.92-0.96 while recall@0.9 is 0.6-0.9.
BF16_CALIBRATED = {'s_640': {'logit_l2': 0.02, 'r90': 0.55, 'r50': 0.93, 'top100_r50': 0.93}, 'l_256': {'logit_l2': 0.02, 'r90': 0.87, 'r50': 0.89, 'top100_r50': 0.93},
                   'l_1024': {'logit_l2': 0.02, 'r90': 0.87, 'r50': 0.90, 'top100_r50': 0.94}}     # l_1024 measured: logits 0.4-0.6 %, 0.920 / 0.943 / 0.980


def _calibrated_model(g, variant, nc, half):
    model = build(variant, nc, synth.make_hyp(conf_thres=float(g['conf_thres'])))
    with torch.no_grad():
        for conv, sc in zip(model.headers['det'].m, g['det_scales']):
            conv.weight.div_(float(sc))
    model.eval()
    return model.half() if half else model


@pytest.mark.parametrize('tag,variant', [('s_640', 's'), ('l_256', 'l'), ('l_1024', 'l')])     # l_1024: BASELINE configs[3]'s tile (64 512 candidates), one tile of its batch
def test_calibrated_eval_detections_fp32_and_bf16_against_reference_golden(golden_dir, tag, variant):
    """The reference's fp32 detections on logit-calibrated weights (eval_cal_*.npz) against (a) the HIP fp32 path: logits to 1e-4, boxes as the other
    full-size goldens; (b) `model.half().eval()` (val_nuclei.py:116,143), the bf16 kernels of the benchmark (dispatch log asserted): recall of the
    reference's detections at IoU >= 0.9 with equal labels."""
    from hd_yolo_amd import _lib
    g = np.load(os.path.join(golden_dir, f'eval_cal_{tag}.npz'))
    batch, size, nc = (int(v) for v in g['meta'])
    x = synth.synth_images(batch, size, seed=7).to(DEV)
    # (a) fp32
    model = _calibrated_model(g, variant, nc, half=False)
    with torch.no_grad():
        _, outputs = model(x)
        plan = list(model._eng().plans.values())[-1]
        for i, d in enumerate(plan.det_views()):
            d64 = d.double()
            got = np.array([d64.sum().item(), d64.abs().sum().item(), d64.pow(2).sum().sqrt().item()])
            assert abs(got[2] - g[f'det_{i}_sums'][2]) <= 1e-4 * g[f'det_{i}_sums'][2], (i, got, g[f'det_{i}_sums'])
    for b in range(batch):
        rb, gb = g[f'out_{b}_boxes'], outputs[b]['det']['boxes'].cpu().numpy()
        iou = _iou_matrix(rb, gb)
        assert (iou.max(1) >= 0.999).mean() >= 0.99, f'fp32 detections of tile {b} differ from the reference'
    # (b) bf16, on the kernels the benchmark runs
    model = _calibrated_model(g, variant, nc, half=True)
    with _lib.option('HDY_DEEP_MIN_TILES', 1):
        _lib.dispatch_log(reset=True)
        with torch.no_grad():
            _, outputs = model(x)
        log = set(_lib.dispatch_log(reset=True))
    assert any(n.startswith('deep_256x') for n in log), sorted(log)
    if variant == 's':
        assert 'conv3x3_c64' in log, sorted(log)
    # the bf16 network's logits against the reference's (the fixture's strip: first tile, first anchor, four grid rows of every level): the error that
    # every detection-level difference below comes from
    plan = list(model._eng().plans.values())[-1]
    errs = []
    for i, d in enumerate(plan.det_views()):
        ref = torch.from_numpy(g[f'det_{i}_strip'])
        got = d[:1, :1, :4].float().cpu()
        errs.append(float((got - ref).norm() / ref.norm()))
    r90, r50, top50, ious = [], [], [], []
    for b in range(batch):
        o = outputs[b]['det']
        rb, rl, rs = g[f'out_{b}_boxes'], g[f'out_{b}_labels'], g[f'out_{b}_scores']
        gb, gl = o['boxes'].float().cpu().numpy(), o['labels'].cpu().numpy()
        assert len(gb) > 0 and np.isfinite(gb).all()
        iou = _iou_matrix(rb, gb)
        iou[rl[:, None] != gl[None, :]] = 0.0
        best = iou.max(1)
        r90.append(float((best >= 0.9).mean()))
        r50.append(float((best >= 0.5).mean()))
        top = np.argsort(-rs.reshape(-1))[:100]                 # the reference's 100 highest-scoring detections: far from the 300th-place cut
        top50.append(float((best[top] >= 0.5).mean()))
        ious.append(float(best.mean()))
    m90, m50, mtop = float(np.mean(r90)), float(np.mean(r50)), float(np.mean(top50))
    print(f'bf16 detections vs reference fp32 goldens, calibrated logits [{tag}]: logits rel. L2 error per level {[round(e, 4) for e in errs]}; '
          f'recall@IoU0.9 {m90:.3f} {[round(f, 3) for f in r90]}, recall@IoU0.5 {m50:.3f}, top-100 recall@IoU0.5 {mtop:.3f}, mean best IoU {np.mean(ious):.3f}, '
          f'kept {[len(outputs[b]["det"]["boxes"]) for b in range(batch)]}')
    lim = BF16_CALIBRATED[tag]
    assert max(errs) <= lim['logit_l2'], (errs, lim)
    assert m90 >= lim['r90'] and m50 >= lim['r50'] and mtop >= lim['top100_r50'], (m90, m50, mtop, lim)


def test_loss_trajectory_fp32_follows_the_reference_and_bf16_stays_in_its_band(golden_dir):
    """30 optimizer steps on one fixed batch (yolov5n, 2 classes, 4 tiles of 128 x 128): the reference's own losses (trajectory_n_128.npz: its modules,
    torch.optim.SGD with train.py's three parameter groups) against the HIP path in fp32 — step by step, the tolerance growing with the step because
    thirty updates amplify fp32 reduction-order noise — and against the bf16 path.  The loop is CHAOTIC (train-mode BatchNorm over four tiles, random
    weights): the reference deviates from ITSELF by up to 4.4 % with 8 host threads instead of 1 and by up to 5.7 % when every weight is perturbed by
    1e-6 relative (both trajectories are in the fixture; a perturbation grows ~10x per step over the first four steps).  So: the first four steps are held
    to 2e-4 x 10^step, every later step to TWICE the reference's own largest deviation (~11 %), and the bf16 trajectory to the same band around the fp32
    one — measured on MI355X: fp32 against the reference 3.7 % at most, bf16 against fp32 6.5 % at most, 5.9 % at step 30: bf16 storage moves this
    training loop no further than a 1e-6 weight perturbation moves the reference (single gradients at cosine ~0.9 notwithstanding).
    Both must also have trained: final loss under a fifth of the first."""
    from hd_yolo_amd.optim import SGD
    g = np.load(os.path.join(golden_dir, 'trajectory_n_128.npz'))
    batch, size, nc, nmin, nmax, steps = (int(v) for v in g['meta'])
    ref = g['losses']
    runs = {}
    for half in (False, True):
        model = build('n', nc).train()
        if half:
            model.half()
        x = synth.synth_images(batch, size, seed=11).to(DEV)
        targets = synth.synth_targets(batch, size, nc, nmin=nmin, nmax=nmax, seed=5)
        g_bn, g_w, g_b = [], [], []
        for m in model.modules():
            if hasattr(m, 'bias') and isinstance(m.bias, torch.nn.Parameter):
                g_b.append(m.bias)
            if isinstance(m, torch.nn.BatchNorm2d):
                g_bn.append(m.weight)
            elif hasattr(m, 'weight') and isinstance(m.weight, torch.nn.Parameter):
                g_w.append(m.weight)
        opt = SGD(g_bn, lr=float(g['lr']), momentum=float(g['momentum']), nesterov=True)
        opt.add_param_group({'params': g_w, 'weight_decay': float(g['weight_decay'])})
        opt.add_param_group({'params': g_b})
        losses = []
        for _ in range(steps):
            out, _ = model(x, targets)
            loss = out['det']['det_loss']
            loss.backward()
            opt.step()
            opt.zero_grad(set_to_none=True)
            losses.append(float(loss.detach()))
        runs[half] = np.array(losses)
    rel32 = np.abs(runs[False] - ref) / ref
    rel16 = np.abs(runs[True] - runs[False]) / runs[False]
    print('loss trajectory: reference first / last', ref[0], ref[-1], '| fp32 max rel', rel32.max(), 'at step', int(rel32.argmax()),
          '| bf16 vs fp32 max rel', rel16.max(), 'final', rel16[-1])
    own = max(float((np.abs(g['losses_8_threads'] - ref) / ref).max()), float((np.abs(g['losses_perturbed_1e6'] - ref) / ref).max()))
    band = 2.0 * own
    assert 0.02 < own < 0.10, own                                 # the fixture's band is what the docstring says it is
    assert (rel32[:4] < 2e-4 * 10.0 ** np.arange(4)).all(), rel32[:4]
    assert rel32.max() < band and rel16.max() < band, (rel32.max(), rel16.max(), band)
    assert runs[True][-1] < 0.2 * runs[True][0] and runs[False][-1] < 0.2 * runs[False][0]


def test_backbone_neck_head_called_separately_match_reference(golden_dir):
    """CSPDarkNet.forward, FPN.forward (bare feature dict, mutated like the reference's) and Detect.forward as stand-alone calls
    (reference: yolov5.py:47-77, yolo_head.py:132-183) against the same goldens as the whole-model run."""
    g = np.load(os.path.join(golden_dir, 'stages_n_64.npz'))
    batch, size, nc = (int(v) for v in g['meta'])
    model = build('n', nc, synth.make_hyp(conf_thres=float(g['conf_thres']))).eval()
    x = synth.synth_images(batch, size, seed=7).to(DEV)
    with torch.no_grad():
        feats = model.backbone(x)
        assert sorted(feats) == sorted(model.backbone.save)
        for k, v in feats.items():
            assert relmax(v, g[f'backbone_{k}']) < 1e-4
        fed = {k: v.clone() for k, v in feats.items()}
        neck = model.neck(fed)
        assert sorted(neck) == sorted(model.neck.save) and -1 in fed and all(k in fed for k in model.neck.save)
        for k, v in neck.items():
            assert relmax(v, g[f'neck_{k}']) < 1e-4
        losses, outputs = model.headers['det'](neck)
    assert losses == {}
    for b in range(batch):
        o = outputs[b]
        np.testing.assert_allclose(o['boxes'].cpu().numpy(), g[f'out_{b}_boxes'], rtol=1e-4, atol=1e-3)
        assert np.array_equal(o['labels'].cpu().numpy(), g[f'out_{b}_labels'])
    model.train()
    with pytest.raises(RuntimeError, match='forward-only'):
        model.neck({k: v.clone() for k, v in feats.items()})


def test_eval_nms_order_bit_exact_vs_oracle():
    """Same logits -> decode -> NMS on both sides: kept rows must be identical and in identical order."""
    from oracle.ref_net import RefNet
    nc = 8
    model = build('s', nc, synth.make_hyp(conf_thres=0.02)).eval()
    x = synth.synth_images(2, 128, seed=3).to(DEV)
    with torch.no_grad():
        model(x)
        plan = next(iter(model._eng().plans.values()))
        dets = [d.contiguous() for d in plan.det_views()]
        head = model.headers['det']
        flat = head.decode_all(dets)
        from metayolo.models.utils_general import nms_per_image
        got = nms_per_image(flat, nc, 0.02, 0.45, 300)
    net = RefNet(synth.make_cfg('s', nc), synth.make_hyp(conf_thres=0.02))
    # feed the oracle the GPU's decoded rows so that the comparison isolates filter + sort + suppression
    from oracle import nms_ref
    ref = nms_ref.nms_per_image_numpy(flat.cpu().numpy(), nc, 0.02, 0.45, 300)
    assert sum(len(r['index']) for r in ref) > 20
    for b in range(2):
        assert np.array_equal(got[b]['index'].cpu().numpy(), ref[b]['index'])
        assert np.array_equal(got[b]['boxes'].cpu().numpy(), ref[b]['boxes'])
    # and the oracle's own decode of the same logits agrees with the kernel's
    opreds = net.decode([d.cpu() for d in dets])
    oflat = torch.cat([torch.nn.functional.pad(p.reshape(2, -1, nc + 5), [0, 1], value=float(i)) for i, p in enumerate(opreds)], 1)
    np.testing.assert_allclose(flat.cpu().numpy(), oflat.numpy(), rtol=1e-4, atol=1e-3)


def test_fuse_keeps_eval_outputs(golden_dir):
    g = np.load(os.path.join(golden_dir, 'stages_n_64.npz'))
    batch, size, nc = (int(v) for v in g['meta'])
    model = build('n', nc).eval()
    x = synth.synth_images(batch, size, seed=7).to(DEV)
    with torch.no_grad():
        model.fuse()
        assert not any(hasattr(m, 'bn') for m in model.modules() if type(m).__name__ == 'Conv')
        feats = model.features(x)
        for k in g.files:
            if k.startswith('fused_neck_'):
                assert relmax(feats[int(k[11:])], g[k]) < 2e-4


@pytest.mark.parametrize('fused', ['1', '0'])
@pytest.mark.parametrize('tag,variant', [('n_64', 'n'), ('s_128', 's'), ('n6_128', 'n6'), ('n_64_ragged', 'n'), ('c1_640', 'n'), ('m_128', 'm'), ('m_640', 'm')])     # m_640: BASELINE configs[2]'s tile geometry (two tiles of its per-GPU batch)
def test_train_step_matches_reference_golden_fp32(golden_dir, tag, variant, fused, monkeypatch):
    """fused = '1': target assignment + loss + logits gradient by csrc/loss.hip; '0': the tensor-expression DetLoss."""
    monkeypatch.setenv('HDY_FUSED_LOSS', fused)
    g = np.load(os.path.join(golden_dir, f'train_{tag}.npz'))
    batch, size, nc, nmin, nmax = (int(v) for v in g['meta'])
    model = build(variant, nc).train()
    x = synth.synth_images(batch, size, seed=11).to(DEV)
    targets = ragged(synth.synth_targets(batch, size, nc, nmin=nmin, nmax=nmax, seed=5), tag)
    losses, outputs = model(x, targets, compute_masks=True)
    loss = losses['det']['det_loss'] + losses['det']['mask_loss']
    loss.backward()
    rtol = 2e-4
    np.testing.assert_allclose(losses['det']['det_loss'].detach().cpu().numpy(), g['loss'], rtol=rtol)
    for k in ('box', 'obj', 'cls'):
        np.testing.assert_allclose(losses['det']['loss_items'][k].cpu().numpy(), g[f'loss_{k}'], rtol=rtol)
    sd = model.state_dict()
    params = dict(model.named_parameters())
    for k in g.files:
        if k.startswith('stat:'):
            assert relmax(sd[k[5:]], g[k]) < 1e-4, k
        elif k.startswith('grad:'):
            assert relmax(params[k[5:]].grad, g[k]) < 1e-3, k
    bad = []
    for name, (s, a, l2) in zip(g['gradsum_names'], g['gradsum']):
        gr = params[str(name)].grad
        assert gr is not None, name
        got = gr.double().pow(2).sum().sqrt().item()
        if abs(got - l2) > 2e-3 * l2 + 1e-9:
            bad.append((str(name), got, l2))
    assert not bad, bad[:8]
    assert int(sd['backbone.0.bn.num_batches_tracked']) == 1
    # a second backward without zero_grad accumulates (train.py's `accumulate` micro-steps)
    g1 = params['backbone.1.conv.weight'].grad.clone()
    losses, _ = model(x, ragged(synth.synth_targets(batch, size, nc, nmin=nmin, nmax=nmax, seed=5), tag), compute_masks=True)
    losses['det']['det_loss'].backward()
    g2 = params['backbone.1.conv.weight'].grad
    assert relmax(g2 - g1, g1) < 0.5 and (g2 - g1).abs().max() > 0      # second step has different BN statistics: not 2x, but added


def test_freeze_backbone_matches_oracle():
    """Model.freeze(['backbone']) (reference: yolo.py:103-107, utils_torch.py:163-203): frozen parameters get no gradient, their
    BatchNorms use (and keep) the running statistics in training mode, nothing is back-propagated below the first trainable
    layer; the trainable neck / head gradients and the loss follow the oracle with the same layers frozen."""
    from oracle.ref_net import RefNet
    nc, B, S = 2, 2, 64
    cfg, hyp = synth.make_cfg('n', nc), synth.make_hyp()
    model = build('n', nc)
    model.freeze(['backbone', 'neck.0'])
    model.train()
    assert type(model.backbone[0].bn).__name__ == 'FrozenBatchNorm2d' and 'backbone.0.bn.num_batches_tracked' not in model.state_dict()
    assert not model.backbone[2].cv1.conv.weight.requires_grad and model.neck[3].cv1.conv.weight.requires_grad
    net = RefNet(cfg, hyp)
    net.frozen = ('backbone', 'neck.0')
    sd = net.init_state()
    train_keys = [k for k in sd if 'running' not in k and not k.startswith('backbone.') and not k.startswith('neck.0.')]
    for k in train_keys:
        sd[k].requires_grad_(True)
    x = synth.synth_images(B, S, seed=11)
    targets = synth.synth_targets(B, S, nc, nmin=3, nmax=8, seed=5)
    rm0 = sd['backbone.1.bn.running_mean'].clone()
    loss_c, _, _ = net.train_forward(sd, x, synth.synth_targets(B, S, nc, nmin=3, nmax=8, seed=5))
    loss_c.backward()
    assert torch.equal(sd['backbone.1.bn.running_mean'], rm0)
    losses, _ = model(x.to(DEV), targets)
    losses['det']['det_loss'].backward()
    assert abs(float(losses['det']['det_loss']) - float(loss_c)) < 2e-4 * abs(float(loss_c))
    params = dict(model.named_parameters())
    assert all(p.grad is None for k, p in params.items() if k.startswith('backbone.') or k.startswith('neck.0.'))
    assert relmax(model.state_dict()['backbone.1.bn.running_mean'], rm0) < 1e-6
    worst = max(relmax(params[k].grad, sd[k].grad) for k in train_keys)
    assert worst < 2e-3, worst
    # the backward list of the frozen part is gone: no weight gradient / dgrad launches for backbone layers
    plan = next(p for p in model._eng().plans.values() if p.training)
    n_wgrad = sum(1 for r in plan.bwd if r[0] == '@fork')
    n_conv = sum(1 for u in plan.units if type(u).__name__ in ('ConvUnit', 'DetUnit'))
    assert n_wgrad < n_conv - 20


def test_three_sgd_steps_track_the_oracle():
    """fwd + bwd + SGD(nesterov) x3 in fp32: the loss trajectory and the updated weights follow the CPU oracle."""
    from oracle.ref_net import RefNet
    nc, B, S = 2, 2, 64
    cfg, hyp = synth.make_cfg('n', nc), synth.make_hyp()
    model = build('n', nc).train()
    net = RefNet(cfg, hyp)
    sd = net.init_state()
    names = [k for k in sd if 'running' not in k]
    for k in names:
        sd[k].requires_grad_(True)
    params = dict(model.named_parameters())
    opt_g = torch.optim.SGD([params[k] for k in names], lr=0.002, momentum=0.9, nesterov=True)
    opt_c = torch.optim.SGD([sd[k] for k in names], lr=0.002, momentum=0.9, nesterov=True)
    before = {k: sd[k].detach().clone() for k in names}
    x = synth.synth_images(B, S, seed=11)
    xg = x.to(DEV)
    for step in range(3):
        lg, _ = model(xg, synth.synth_targets(B, S, nc, nmin=3, nmax=8, seed=5))
        lg['det']['det_loss'].backward()
        opt_g.step()
        opt_g.zero_grad(set_to_none=True)
        lc, _, _ = net.train_forward(sd, x, synth.synth_targets(B, S, nc, nmin=3, nmax=8, seed=5))
        lc.backward()
        opt_c.step()
        opt_c.zero_grad(set_to_none=True)
        a, b = lg['det']['det_loss'].item(), lc.item()
        # the tiny 64x64 / batch-2 case normalises over as few as 8 samples per channel: trajectories separate quickly,
        # so the step size is small and the bar is on the UPDATE (delta of the weights), not on long-run agreement
        assert abs(a - b) <= 5e-3 * abs(b), (step, a, b)
    for k in ('backbone.0.conv.weight', 'backbone.4.cv3.conv.weight', 'neck.13.cv3.bn.weight', 'headers.det.m.1.bias'):
        dg, dc = params[k].detach().cpu() - before[k], sd[k].detach() - before[k]
        assert dc.abs().max() > 0 and relmax(dg, dc) < 3e-2, (k, relmax(dg, dc))
    msd = model.state_dict()
    assert relmax(msd['backbone.2.cv1.bn.running_var'], sd['backbone.2.cv1.bn.running_var']) < 1e-3
    assert int(msd['backbone.2.cv1.bn.num_batches_tracked']) == 3


def test_train_step_bf16_is_close_to_fp32():
    nc = 8
    model = build('s', nc).train()
    x = synth.synth_images(4, 256, seed=11).to(DEV)
    t1 = synth.synth_targets(4, 256, nc, nmin=20, nmax=60, seed=5)
    t2 = synth.synth_targets(4, 256, nc, nmin=20, nmax=60, seed=5)
    ref_model = build('s', nc).train()
    l32, _ = ref_model(x, t1)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        l16, _ = model(x, t2)
    a, b = l32['det']['det_loss'].item(), l16['det']['det_loss'].item()
    assert abs(a - b) / a < 0.03, (a, b)
    l32['det']['det_loss'].backward()
    l16['det']['det_loss'].backward()
    p32, p16 = dict(ref_model.named_parameters()), dict(model.named_parameters())
    # This randomly initialised train-mode-BN network is chaotic: rounding only the WEIGHTS and the input to bf16 inside the
    # exact fp32 pipeline already moves deep-layer gradients to cosine 0.86-0.96 (scripts/bf16_grad_check.py, measured on
    # MI355X).  So the bf16 plan is held to: the shallowest gradient path (head) agrees to 1e-3, and the bulk of the layers
    # stay strongly aligned.  Per-kernel bf16 exactness is what tests/test_gpu_kernels.py pins.
    cos = {}
    for k in p32:
        u, v = p32[k].grad.flatten().double(), p16[k].grad.flatten().double()
        cos[k] = (torch.dot(u, v) / (u.norm() * v.norm() + 1e-30)).item()
    assert cos['headers.det.m.0.weight'] > 0.999 and cos['headers.det.m.2.bias'] > 0.999, cos
    vals = sorted(cos.values())
    assert vals[len(vals) // 2] > 0.85 and vals[0] > 0.6, (vals[0], vals[len(vals) // 2])


def test_cpu_tensors_fail_loudly():
    from hd_yolo_amd._lib import HdyError
    from metayolo.models.yolo import Model
    model = Model(synth.make_cfg('n', 2), synth.make_hyp()).eval()
    with pytest.raises(HdyError):
        model(torch.zeros(1, 3, 64, 64))


def test_bf16_training_on_a_fixed_batch_reduces_the_loss():
    """End-to-end sanity of the bf16 step (HIP forward + fused DetLoss + backward + SGD on fp32 masters): on one fixed batch the
    loss must fall steadily and stay finite — a sign error or a dropped gradient anywhere in the backward list shows up here."""
    nc = 8
    model = build('s', nc).train().half()
    x = synth.synth_images(8, 128, seed=3).to(DEV)
    targets = synth.synth_targets(8, 128, nc, nmin=5, nmax=15, seed=4)
    opt = torch.optim.SGD(model.parameters(), lr=0.005, momentum=0.9, nesterov=True)
    losses = []
    for _ in range(40):
        out, _ = model(x, targets)
        loss = out['det']['det_loss']
        loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        losses.append(float(loss.detach()))
    assert all(l == l and abs(l) < 1e6 for l in losses), losses
    assert losses[-1] < 0.7 * losses[0], (losses[0], losses[-1])
    assert min(losses[-5:]) < min(losses[:5])


def test_eval_plan_follows_weight_updates():
    """Inference plans skip the weight re-pack and the folded BatchNorm coefficients while the version counters of their sources stand
    still; an in-place update of a weight or of a running statistic (what optimizers, load_state_dict and EMA do) must show."""
    m = build('n', 2).eval()
    x = synth.synth_images(1, 64, seed=7).to(DEV)
    with torch.no_grad():
        plan = m._eng().plan_for(x, False, torch.float32)
        plan.run_forward(x)
        a = plan.feature(9).clone()
        plan.run_forward(x)
        assert torch.equal(plan.feature(9), a)
        m.backbone[0].conv.weight.mul_(1.5)
        plan.run_forward(x)
        b = plan.feature(9).clone()
        assert not torch.equal(a, b)
        m.backbone[1].bn.running_var.mul_(4.0)
        plan.run_forward(x)
        assert not torch.equal(plan.feature(9), b)


def _empty_like(t):
    a = t['anns']['det'][0]
    return {'anns': {'det': [{'boxes': a['boxes'][:0].clone(), 'labels': a['labels'][:0].clone()}]}}


@pytest.mark.parametrize('which', ['one image without targets', 'no targets at all'])
@pytest.mark.parametrize('fused', ['1', '0'])
def test_images_without_targets_match_oracle(which, fused, monkeypatch):
    """Ragged / empty ground truth (the reference handles a tile without nuclei: datasets.py:462-519 emits empty boxes): the loss is
    the objectness term alone where nothing matches, and it must be the oracle's, gradients included."""
    from oracle.ref_net import RefNet
    monkeypatch.setenv('HDY_FUSED_LOSS', fused)
    nc, B, S = 2, 2, 64
    cfg, hyp = synth.make_cfg('n', nc), synth.make_hyp()
    model = build('n', nc).train()
    net = RefNet(cfg, hyp)
    sd = net.init_state()
    names = [k for k in sd if 'running' not in k]
    for k in names:
        sd[k].requires_grad_(True)
    x = synth.synth_images(B, S, seed=11)

    def make():
        t = synth.synth_targets(B, S, nc, nmin=3, nmax=8, seed=5)
        return [_empty_like(t[0]), t[1]] if which.startswith('one') else [_empty_like(q) for q in t]

    lg, _ = model(x.to(DEV), make())
    lg['det']['det_loss'].backward()
    lc, items, _ = net.train_forward(sd, x, make())
    lc.backward()
    a, b = float(lg['det']['det_loss'].detach()), float(lc.detach())
    assert a == a and abs(a - b) <= 2e-4 * abs(b), (a, b)
    if not which.startswith('one'):
        assert float(lg['det']['loss_items']['box']) == 0.0 and float(lg['det']['loss_items']['cls']) == 0.0
    params = dict(model.named_parameters())
    for k in ('backbone.0.conv.weight', 'neck.13.cv3.conv.weight', 'headers.det.m.0.bias', 'headers.det.m.2.weight'):
        assert relmax(params[k].grad, sd[k].grad) < 2e-3, k


def test_no_detection_above_threshold_gives_empty_outputs():
    m = build('n', 2, synth.make_hyp(conf_thres=0.9999)).eval()
    with torch.no_grad():
        _, out = m(synth.synth_images(3, 64, seed=7).to(DEV))
    assert len(out) == 3
    for o in out:
        assert tuple(o['det']['boxes'].shape) == (0, 4) and tuple(o['det']['scores'].shape) == (0,) and tuple(o['det']['labels'].shape) == (0,)
    with torch.no_grad():
        _, one = m(synth.synth_images(1, 64, seed=7).to(DEV))            # batch of one tile
    assert len(one) == 1


def test_non_square_tiles_match_oracle():
    """Rectangular tiles (96 x 160): every level's grid is 3:5, the NHWC pitches and the decode's (ny, nx) differ — eval detections,
    training loss and gradients against the oracle on the same tensors (normalised targets do not depend on the shape)."""
    from oracle.ref_net import RefNet
    nc, B, H, W = 2, 2, 96, 160
    cfg, hyp = synth.make_cfg('n', nc), synth.make_hyp(conf_thres=0.05)
    g = torch.Generator().manual_seed(77)
    x = torch.rand((B, 3, H, W), generator=g)
    net = RefNet(cfg, hyp)
    sd = net.init_state()
    model = build('n', nc, hyp).eval()
    with torch.no_grad():
        _, out = model(x.to(DEV))
        _, _, _, ref = net.eval_forward(sd, x)
    for o, r in zip(out, ref):
        assert o['det']['boxes'].shape == r['boxes'].shape and len(r['boxes']) > 0
        np.testing.assert_allclose(o['det']['boxes'].cpu().numpy(), r['boxes'].numpy(), rtol=1e-4, atol=1e-3)
        assert np.array_equal(o['det']['labels'].cpu().numpy(), r['labels'].numpy())
    model.train()
    names = [k for k in sd if 'running' not in k]
    for k in names:
        sd[k].requires_grad_(True)
    t = lambda: synth.synth_targets(B, 128, nc, nmin=3, nmax=8, seed=5)
    lg, _ = model(x.to(DEV), t())
    lg['det']['det_loss'].backward()
    lc, _, _ = net.train_forward(sd, x, t())
    lc.backward()
    a, b = float(lg['det']['det_loss'].detach()), float(lc.detach())
    assert abs(a - b) <= 2e-4 * abs(b), (a, b)
    params = dict(model.named_parameters())
    for k in ('backbone.0.conv.weight', 'backbone.4.cv3.conv.weight', 'neck.13.cv3.conv.weight', 'headers.det.m.1.weight'):
        assert relmax(params[k].grad, sd[k].grad) < 2e-3, k


@pytest.mark.parametrize('variant', ['s', 'm'])
def test_bf16_plan_layer_by_layer_against_fp32_torch(variant):
    """The bf16 training plan (every conv / BatchNorm / SiLU / residual / fused-backward launch, all pitches and channel slices) checked
    LOCALLY: each unit's outputs are recomputed in fp32 torch from that unit's OWN bf16 inputs as the plan left them in its buffers —
    forward (raw conv output, BatchNorm coefficients, activation) and backward (weight gradient, BatchNorm parameter gradients, and the
    data gradient where the unit is the tensor's only consumer).  Whole-network comparisons cannot do this: this random-init
    train-mode-BN network amplifies a bf16 rounding of the weights alone to gradient cosines of 0.86 (scripts/bf16_grad_check.py), so a
    wrong pitch on one bf16-only path would hide in the noise; per unit the agreement is at the bf16 rounding level (2^-7 .. 2^-6 of max)."""
    import torch.nn.functional as F
    from hd_yolo_amd import plan as planmod
    nc, B, S = 8, 4, 128
    model = build(variant, nc).train()          # 'm': BASELINE config C3's widths (48 / 96 / 192 / 384 / 768 channels, 16-byte rows that are not 128-byte multiples)
    model.half()
    x = synth.synth_images(B, S, seed=11).to(DEV)
    losses, _ = model(x, synth.synth_targets(B, S, nc, nmin=10, nmax=30, seed=5))
    losses['det']['det_loss'].backward()
    torch.cuda.synchronize()
    eng = model._eng()
    plan = next(iter(eng.plans.values()))
    assert plan.dtype == torch.bfloat16
    units = [u for u in plan.units if isinstance(u, planmod.ConvUnit)]
    if variant == 's':
        assert sum(plan._fusable_1x1(u) for u in units) >= 8, 'the fused 1x1 backward must be part of what is checked'
    if os.environ.get('HDY_EXPECT_PRODUCER_STATS') and variant == 's':
        assert plan.producer_stat_units >= 10, plan.producer_stat_units

    def consumers(v):
        n = 0
        for u in plan.units:
            if isinstance(u, planmod.ConvUnit):
                n += (u.x is v) + (u.res is v)
            elif isinstance(u, (planmod.PoolUnit, planmod.UpUnit, planmod.DetUnit)):
                n += u.x is v
        return n + (consumers(v.cat) if v.cat is not None else 0)

    q = lambda t: t.to(torch.bfloat16).float()
    nchw = lambda t: t.float().permute(0, 3, 1, 2)
    worst = {}

    def note(kind, name, got, ref, tol, l2=False):
        e = ((got - ref).norm() / (ref.norm() + 1e-30)).item() if l2 else relmax(got, ref.detach().cpu().numpy())
        worst[kind] = max(worst.get(kind, 0.0), e)
        assert e < tol, f'{kind} of {name}: {e:.3e} >= {tol}'

    checked_dx = 0
    for ui, u in enumerate(units):
        name = f'unit {ui} ({u.C}->{u.K} k{u.k} s{u.s})'
        w = torch.cat([m.conv.weight for m in u.mods]).detach()
        if u.stem:
            xin = q(x)
        else:
            xin = nchw(u.x.t())
        y_ref = F.conv2d(xin, q(w), None, u.s, u.p)
        note('raw conv output', name, nchw(u.yraw), y_ref, 6e-3)            # one bf16 rounding of the output: 2^-8 of the element, measured 3.5e-3 of max
        # BatchNorm coefficients from the fp32 accumulators' statistics
        mean, var = y_ref.mean((0, 2, 3)), y_ref.var((0, 2, 3), unbiased=False)
        gamma, beta = torch.cat([m.bn.weight for m in u.mods]).detach(), torch.cat([m.bn.bias for m in u.mods]).detach()
        invstd = 1.0 / torch.sqrt(var + u.mods[0].bn.eps)
        note('BN scale', name, u.scale, gamma * invstd, 2e-3)
        note('BN shift', name, u.shift, beta - mean * gamma * invstd, 5e-3)
        # activation from the plan's own raw output and coefficients
        yr = nchw(u.yraw)
        z_ref = F.silu(yr * u.scale.view(1, -1, 1, 1) + u.shift.view(1, -1, 1, 1)) if u.act == 1 else yr * u.scale.view(1, -1, 1, 1) + u.shift.view(1, -1, 1, 1)
        if u.res is not None:
            z_ref = z_ref + nchw(u.res.t())
        k0 = 0
        for o in u.outs:
            note('activation', name, nchw(o.t()), z_ref[:, k0:k0 + o.c], 6e-3)
            k0 += o.c
        # ---- backward of this unit from the gradient the plan holds for its outputs.  A Bottleneck's shortcut input shares gradient
        # storage with the block output (plan.Val.galias): after the whole backward that storage holds the INPUT's gradient, so the
        # output gradient this unit saw is gone — its backward is covered by the kernel tests and by the fp32 goldens
        if u.res is not None:
            continue
        dz = torch.cat([nchw(o.gread()) for o in u.outs], 1)
        uu = yr * u.scale.view(1, -1, 1, 1) + u.shift.view(1, -1, 1, 1)
        sg = torch.sigmoid(uu)
        du = dz * (sg * (1 + uu * (1 - sg))) if u.act == 1 else dz
        xh = (yr - u.mean.view(1, -1, 1, 1)) * u.invstd.view(1, -1, 1, 1)
        M = yr.shape[0] * yr.shape[2] * yr.shape[3]
        dbeta, dgamma = du.sum((0, 2, 3)), (du * xh).sum((0, 2, 3))
        dy = q(u.scale.view(1, -1, 1, 1) * (du - (dbeta / M).view(1, -1, 1, 1) - xh * (dgamma / M).view(1, -1, 1, 1)))
        k0 = 0
        for m in u.mods:
            K = m.conv.out_channels
            note('dgamma', name, m.bn.weight.grad, dgamma[k0:k0 + K], 1e-4, l2=True)
            note('dbeta', name, m.bn.bias.grad, dbeta[k0:k0 + K], 1e-4, l2=True)
            gw = torch.nn.grad.conv2d_weight(xin, m.conv.weight.shape, dy[:, k0:k0 + K].contiguous(), u.s, u.p)
            note('weight gradient', name, m.conv.weight.grad, gw, 1e-3, l2=True)      # same bf16 dy on both sides: fp32 summation order only (measured 6e-5)
            k0 += K
        aliased = {id(v.galias) for v in plan.vals if v.galias is not None}       # storage later overwritten with a shortcut input's gradient
        members = [u.x] + ([pv for pv, _ in u.x.parts] if (not u.stem and u.x.parts is not None) else [])
        if not u.stem and u.x.needs_grad and u.x.galias is None and not any(id(v) in aliased for v in members):
            xv = u.x
            single = consumers(xv) == 1 if xv.parts is None else (consumers(xv) == 1 and all(consumers(pv) == 1 for pv, _ in xv.parts))
            if single:
                dx = torch.nn.grad.conv2d_input(xin.shape, q(w), dy, u.s, u.p)
                note('data gradient', name, nchw(xv.g()), dx, 6e-3, l2=True)              # + one bf16 rounding of dx (measured 1.7e-3)
                checked_dx += 1
    assert checked_dx >= 10, checked_dx
    print('bf16 layerwise worst errors:', {k: f'{v:.2e}' for k, v in worst.items()})


def test_bf16_plan_layer_by_layer_with_producer_side_statistics():
    """The alternative backward in which the launch that completes a gradient also serves the BatchNorm-backward statistics of the unit(s)
    that gradient belongs to goes through the same unit-by-unit check in all three settings: the default serves statistics from the fused 1x1
    backward kernel only ('fused'), HDY_PRODUCER_STATS=1 from the data-gradient epilogues too (measured slower), 0 switches it off."""
    import subprocess
    import sys
    if os.environ.get('HDY_PRODUCER_STATS_CHILD'):
        return
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for mode, expect in (('1', '1'), ('0', '')):
        env = dict(os.environ, HDY_PRODUCER_STATS=mode, HDY_PRODUCER_STATS_CHILD='1', HDY_EXPECT_PRODUCER_STATS=expect)
        p = subprocess.run([sys.executable, '-m', 'pytest', 'tests/test_gpu_model.py', '-q', '-x', '-k', 'test_bf16_plan_layer_by_layer_against_fp32_torch'],
                           cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0 and ' passed' in p.stdout, (mode, p.stdout[-3000:])


def test_gradients_under_every_zero_grad_style():
    """The parameters' .grad are views of the flat buffer the backward pass writes (no copy per step); a caller that keeps them across a
    backward pass — gradient accumulation, zero_grad(set_to_none=False) — must still see torch's semantics."""
    nc, B, S = 2, 2, 64
    model = build('n', nc).train()
    x = synth.synth_images(B, S, seed=11).to(DEV)
    params = [p for p in model.parameters() if p.requires_grad]

    def backward():
        losses, _ = model(x, synth.synth_targets(B, S, nc, nmin=3, nmax=8, seed=5))
        losses['det']['det_loss'].backward()

    backward()
    g1 = [p.grad.clone() for p in params]
    held = [p.grad for p in params]                       # the tensors a caller may keep
    for p in params:                                      # zero_grad(set_to_none=False)
        p.grad.zero_()
    backward()                                            # same batch statistics, same weights: the same gradient, added to zeros
    for p, g in zip(params, g1):
        assert relmax(p.grad, g) < 1e-4
    backward()                                            # no zeroing: accumulates
    for p, g in zip(params, g1):
        assert relmax(p.grad, 2 * g) < 1e-4
    for p in params:                                      # zero_grad(set_to_none=True)
        p.grad = None
    backward()
    for p, g in zip(params, g1):
        assert relmax(p.grad, g) < 1e-4
    assert all(h.shape == p.shape for h, p in zip(held, params))


@pytest.mark.parametrize('fused', ['1', '0'])
@pytest.mark.parametrize('nc', [1, 5])
def test_class_counts_other_than_the_fixtures_match_oracle(nc, fused, monkeypatch):
    """nc = 1 (a single nucleus class: 6 outputs per anchor, 18 logits padded to 20, no class loss — loss.py:222 `if self.nc > 1`) and an
    odd class count (10 outputs per anchor, 30 -> 32): logits pitch, objectness positions inside the 4-channel groups of the loss
    kernel and the class-gradient slots differ from the 2- and 8-class fixtures."""
    from oracle.ref_net import RefNet
    monkeypatch.setenv('HDY_FUSED_LOSS', fused)
    B, S = 2, 64
    cfg, hyp = synth.make_cfg('n', nc), synth.make_hyp()
    model = build('n', nc).train()
    net = RefNet(cfg, hyp)
    sd = net.init_state()
    for k in sd:
        if 'running' not in k:
            sd[k].requires_grad_(True)
    x = synth.synth_images(B, S, seed=11)
    lg, _ = model(x.to(DEV), synth.synth_targets(B, S, nc, nmin=20, nmax=40, seed=5))       # dense: several matches share a cell
    lg['det']['det_loss'].backward()
    lc, items, _ = net.train_forward(sd, x, synth.synth_targets(B, S, nc, nmin=20, nmax=40, seed=5))
    lc.backward()
    a, b = float(lg['det']['det_loss'].detach()), float(lc.detach())
    assert a == a and abs(a - b) <= 2e-4 * abs(b), (a, b)
    for k in ('box', 'obj', 'cls'):
        assert abs(float(lg['det']['loss_items'][k]) - float(items[k])) <= 2e-4 * abs(float(items[k])) + 1e-7, k
    params = dict(model.named_parameters())
    for k in ('backbone.0.conv.weight', 'neck.13.cv3.conv.weight', 'headers.det.m.0.bias', 'headers.det.m.0.weight', 'headers.det.m.2.weight'):
        assert relmax(params[k].grad, sd[k].grad) < 2e-3, k


def test_multi_scale_sizes_reuse_their_cached_plans():
    """train.py --multi-scale (reference train.py:447-452): the tile size changes between steps; every size gets ONE static plan that is found
    again when the size comes back (no re-trace, same device buffers), and the losses of a size do not depend on which sizes ran in between."""
    nc = 8
    model = build('n', nc).train()
    eng = model._eng()
    seen, losses = {}, {}
    for rnd in range(2):
        for size in (128, 160, 192):
            x = synth.synth_images(2, size, seed=size).to(DEV)
            l, _ = model(x, synth.synth_targets(2, size, nc, nmin=3, nmax=8, seed=size))
            l['det']['det_loss'].backward()
            plan = eng.last_plan
            if rnd == 0:
                seen[size] = plan
            else:
                assert plan is seen[size], f'{size}: the plan was rebuilt'
            losses.setdefault(size, []).append(float(l['det']['det_loss'].detach()))
            for p in model.parameters():
                p.grad = None
    assert len({id(p) for p in seen.values()}) == 3 and sum(1 for p in eng.plans.values() if p.training) == 3
    # train-mode BatchNorm statistics moved between the rounds (running stats only: the batch statistics are the batch's own): same loss
    for size, (a, b) in losses.items():
        assert abs(a - b) < 1e-4 * abs(a), (size, a, b)
