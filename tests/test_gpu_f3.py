"""SURVEY §8 row f3 on the MI355X: torchvision-style NMS on explicit boxes (hdy_nms_boxes), Ensemble.merge against the reference's
output (tests/golden/f3.npz), Deploy, header re-ordering through the launch plans."""
import os

import numpy as np
import pytest
import torch

from hd_yolo_amd import ops, synth
from metayolo.engines.general import manipulate_header_label_order
from metayolo.models.yolo import Deploy, Ensemble, Model
from oracle import nms_ref

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda', 0)
G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'f3.npz'), allow_pickle=False)


@pytest.mark.parametrize('n,iou', [(1, 0.5), (37, 0.45), (1000, 0.3), (5000, 0.6), (9000, 0.45)])
def test_nms_boxes_bit_exact_order(n, iou):
    g = torch.Generator().manual_seed(n)
    c = torch.rand((n, 2), generator=g) * 300
    wh = torch.rand((n, 2), generator=g) * 40 + 4
    boxes = torch.cat([c - wh / 2, c + wh / 2], 1)
    scores = torch.rand(n, generator=g)
    scores[::7] = scores[0]                       # ties: lower index first
    want = nms_ref.nms_numpy(boxes.numpy(), scores.numpy(), iou)
    got = ops.nms(boxes.to(DEV), scores.to(DEV), iou).cpu().numpy()
    np.testing.assert_array_equal(got, want)


def test_nms_boxes_empty():
    assert ops.nms(torch.zeros((0, 4), device=DEV), torch.zeros(0, device=DEV), 0.5).numel() == 0


def test_ensemble_merge_matches_reference():
    ens = Ensemble([], {'conf_thres': 0.3, 'iou_thres': 0.4, 'max_det': 12})
    parts = [{'det': {k: torch.from_numpy(G[f'ens_in_{j}_{k}']).to(DEV) for k in ('boxes', 'scores', 'labels')}} for j in range(3)]
    out = ens.merge(parts)['det']
    for k in ('boxes', 'scores', 'labels'):
        np.testing.assert_array_equal(out[k].cpu().numpy(), G[f'ens_out_{k}'])


def test_ensemble_merge_carries_masks():
    """Masks ride along with their boxes through the score filter and the NMS (reference: yolo.py:172-199); a model without masks
    contributes zero masks for its boxes.  Tagging each mask with its box's position in the concatenation makes the selection visible."""
    ens = Ensemble([], {'conf_thres': 0.3, 'iou_thres': 0.4, 'max_det': 12})
    parts = [{'det': {k: torch.from_numpy(G[f'ens_in_{j}_{k}']).to(DEV) for k in ('boxes', 'scores', 'labels')}} for j in range(3)]
    plain = ens.merge(parts)['det']
    off = 0
    for j, p in enumerate(parts):
        n = len(p['det']['boxes'])
        if j != 1:                                # the middle model has no mask branch
            p['det']['masks'] = (torch.arange(off, off + n, device=DEV, dtype=torch.float32) + 1)[:, None, None, None].expand(n, 1, 28, 28).contiguous()
        off += n
    out = ens.merge(parts)['det']
    for k in ('boxes', 'scores', 'labels'):
        assert torch.equal(out[k], plain[k])
    allb = torch.cat([p['det']['boxes'] for p in parts])
    n0, n1 = len(parts[0]['det']['boxes']), len(parts[1]['det']['boxes'])
    assert out['masks'].shape == (len(out['boxes']), 1, 28, 28)
    for box, m in zip(out['boxes'], out['masks']):
        src = int((allb == box).all(1).nonzero()[0])
        want = 0.0 if n0 <= src < n0 + n1 else float(src + 1)
        assert float(m.min()) == float(m.max()) == want


def _model(variant='n', nc=2):
    m = Model(synth.make_cfg(variant, nc), synth.make_hyp(conf_thres=0.05))
    m.load_state_dict(synth.synth_state_dict(synth.shapes_of(m), seed=0), strict=False)
    return m.to(DEV).eval()


def test_deploy_and_ensemble_forward():
    m = _model()
    x = synth.synth_images(2, 64, seed=7).to(DEV)
    _, want = m(x)
    d = Deploy(m)
    assert sorted(k.split('.')[0] for k in d.state_dict())[0] == 'backbone' and not any(k.startswith('_model') for k in d.state_dict())
    _, got = d(list(x))                           # list of tiles, as the reference's loader hands them over
    for a, b in zip(got, want):
        for k in ('boxes', 'scores', 'labels'):
            assert torch.equal(a['det'][k], b['det'][k])
    # fused copy: same detections within bf16-free fp32 rounding of the folded weights; the original stays unfused
    _, fused = Deploy(m, fuse=True)(x)
    assert hasattr(m.backbone[0], 'bn')
    for a, b in zip(fused, want):
        assert a['det']['boxes'].shape == b['det']['boxes'].shape
        assert torch.allclose(a['det']['boxes'], b['det']['boxes'], rtol=1e-3, atol=1e-2)
    # an ensemble of a model with itself keeps exactly the single model's boxes above the ensemble threshold
    ens = Ensemble([m, m], m.headers['det'].nms_params)
    _, merged = ens(x)
    for a, b in zip(merged, want):
        sel = b['det']['scores'] > ens.nms_params['conf_thres']
        exp = b['det']['boxes'][sel][torch.sort(b['det']['scores'][sel], descending=True, stable=True)[1]]
        assert torch.equal(a['det']['boxes'], exp[:int(ens.nms_params['max_det'])])
        assert (a['det']['scores'][:-1] >= a['det']['scores'][1:]).all()


def test_header_reorder_rebuilds_plans():
    m = _model('n', 3)
    x = synth.synth_images(1, 64, seed=3).to(DEV)
    plan, dets = m._eng().forward(x, False, torch.float32)
    before = [d.clone() for d in dets]
    manipulate_header_label_order(m.headers['det'], [2, 0, 1])
    plan, after = m._eng().forward(x, False, torch.float32)
    for a, b in zip(after, before):               # (bs, na, ny, nx, no): class columns permuted, box/objectness untouched
        assert torch.allclose(a[..., :5], b[..., :5], rtol=1e-5, atol=1e-6)
        assert torch.allclose(a[..., 5:], b[..., 5:][..., [2, 0, 1]], rtol=1e-5, atol=1e-6)


def test_non_max_suppression_every_option_matches_reference():
    """Class-aware entry point with all options against the reference's outputs (tests/golden/nms_options.npz)."""
    from metayolo.models.utils_general import non_max_suppression
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'nms_options.npz'))
    labels = [torch.from_numpy(g['labels_0']), torch.from_numpy(g['labels_1'])]
    cases = {'default': {}, 'multi': {'multi_label': True}, 'agnostic': {'agnostic': True}, 'classes': {'classes': [0, 2]},
             'multi_agnostic_top5': {'multi_label': True, 'agnostic': True, 'max_det': 5}, 'apriori': {'labels': labels}}
    preds = torch.from_numpy(g['preds']).to(DEV)
    for tag, kw in cases.items():
        res = non_max_suppression(preds.clone(), conf_thres=0.2, iou_thres=0.5, **kw)
        for b, d in enumerate(res):
            np.testing.assert_array_equal(d.cpu().numpy(), g[f'{tag}_{b}'], err_msg=f'{tag} image {b}')


# ---------------------------------------------------------------------------------------------- round 2 additions
def test_nms_boxes_negative_scores_and_more_than_one_launch_of_survivors():
    """torchvision.ops.nms takes scores of any sign and returns EVERY survivor: beyond the 4096 entries of the LDS kept list the kernel keeps
    its list in the workspace (whole-slide merges; round 6: one launch), and ranks negative scores below positive ones."""
    g = torch.Generator().manual_seed(5)
    n = 14000
    c = torch.rand((n, 2), generator=g) * 4000                    # sparse: most boxes survive
    wh = torch.rand((n, 2), generator=g) * 30 + 6
    boxes = torch.cat([c - wh / 2, c + wh / 2], 1)
    scores = torch.rand(n, generator=g) * 2 - 1                   # half of them negative
    scores[::11] = scores[3]
    scores[5] = 0.0
    scores[6] = -0.0
    want = nms_ref.nms_numpy(boxes.numpy(), scores.numpy(), 0.4)
    assert len(want) > 2 * 4096
    got = ops.nms(boxes.to(DEV), scores.to(DEV), 0.4).cpu().numpy()
    np.testing.assert_array_equal(got, want)
    got = ops.nms(boxes.to(DEV), scores.to(DEV), 0.4, max_det=5000).cpu().numpy()
    np.testing.assert_array_equal(got, want[:5000])


@pytest.mark.parametrize('tag,ml', [('single', False), ('multi', True)])
def test_compute_outputs_matches_reference_golden(tag, ml):
    """Detect.compute_outputs on the HIP path (decode kernel -> NMS kernel -> score / label logic) against the reference's own
    outputs for the same logits, incl. multi_label=True (reference: yolo_head.py:301-355; tests/golden/outputs.npz)."""
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'outputs.npz'))
    model = Model(synth.make_cfg('n', 3), synth.make_hyp(conf_thres=0.15, multi_label=ml)).to(DEV).eval()
    head = model.headers['det']
    dets = [torch.from_numpy(g[f'{tag}_det_{i}']).to(DEV) for i in range(3)]
    for preds in (head.compute_proposals(dets), head.decode_all(dets)):          # per-level list and the fused (bs, N, no+1) form
        res = head.compute_outputs(preds, [], compute_masks=False)
        assert len(res) == 2
        for b, o in enumerate(res):
            np.testing.assert_allclose(o['boxes'].cpu().numpy(), g[f'{tag}_out_{b}_boxes'], rtol=1e-4, atol=1e-3)
            np.testing.assert_allclose(o['scores'].cpu().numpy(), g[f'{tag}_out_{b}_scores'], rtol=1e-5, atol=1e-6)
            assert np.array_equal(o['labels'].cpu().numpy(), g[f'{tag}_out_{b}_labels'])
            if ml:
                assert o['labels'].dim() == 2 and o['labels'].shape[1] == 1 + head.nc


def test_evaluation_caller_end_to_end(tmp_path):
    """evaluation.py: checkpoint -> build_model -> Deploy -> timed loop with resize and scale_coords back to the tile frame; the same
    tiles run directly through the model at the network size give the same detections up to the rescale + round."""
    import evaluation
    from metayolo.datasets import SyntheticTiles
    from metayolo.models.utils_general import scale_coords
    ref = Model(synth.make_cfg('n', 3), synth.make_hyp(conf_thres=0.05))
    ref.load_state_dict(synth.synth_state_dict(synth.shapes_of(ref), seed=2), strict=False)
    torch.save({'model': ref.state_dict(), 'ema': None, 'epoch': 3}, tmp_path / 'w.pt')
    model, deployed = evaluation.build_model(str(tmp_path / 'w.pt'), ref_model=ref, half=False,
                                             extra_configs={'headers': {'det': {'nms_params': {'conf_thres': 0.05, 'iou_thres': 0.5, 'max_det': 50}}}})
    assert model.headers['det'].nms_params['max_det'] == 50.0 and not model.training
    loader = SyntheticTiles(3, 96, 3, 2, seed=77)                      # 96 px tiles, network input 128
    results, spi = evaluation.inference_on_loader_yolov5(deployed, loader, DEV, input_size=128)
    assert len(results) == 6 and spi > 0
    k = 0
    for images, _ in SyntheticTiles(3, 96, 3, 2, seed=77):
        x = torch.nn.functional.interpolate(torch.stack(list(images)).to(DEV), size=(128, 128), mode='bilinear', align_corners=False)
        _, outs = model(x)
        for o in outs:
            want = scale_coords((128, 128), o['det']['boxes'].clone(), (96, 96)).round().cpu()
            got = results[k]['det']
            assert got['boxes'].device.type == 'cpu' and torch.equal(got['boxes'], want) and torch.equal(got['labels'], o['det']['labels'].cpu())
            k += 1
    assert sum(len(r['det']['boxes']) for r in results) > 0
    ens, ens_dep = evaluation.attempt_load_model([str(tmp_path / 'w.pt')] * 2, ref_model=ref, half=False)
    assert isinstance(ens_dep, Ensemble) and len(ens_dep) == 2


def test_inference_on_slide_merges_rois():
    """Whole-slide protocol (Detect.merge_outputs / rescale_outputs, yolo_head.py:450-471): without overlap the merged result is exactly
    the per-tile detections shifted by their roi; rescale_outputs scales the boxes; with overlap duplicates are removed by one NMS."""
    import evaluation
    m = Model(synth.make_cfg('n', 3), synth.make_hyp(conf_thres=0.05)).to(DEV).eval()
    m.load_state_dict(synth.synth_state_dict(synth.shapes_of(m), seed=2), strict=False)
    dep = Deploy(m)
    slide = synth.synth_images(1, 256, seed=9)[0].to(DEV)
    assert evaluation.slide_rois(256, 256, 128, 0) == [(0, 0), (128, 0), (0, 128), (128, 128)]
    assert evaluation.slide_rois(300, 128, 128, 32) == [(0, 0), (0, 96), (0, 172)]
    out = evaluation.inference_on_slide(dep, slide, tile=128, overlap=0, batch_size=3, scale=0.5)['det']
    want_b, want_s = [], []
    for (x0, y0) in evaluation.slide_rois(256, 256, 128, 0):
        _, o = dep(slide[None, :, y0:y0 + 128, x0:x0 + 128].contiguous())
        want_b.append(o[0]['det']['boxes'] + torch.tensor([x0, y0, x0, y0], device=DEV, dtype=torch.float32))
        want_s.append(o[0]['det']['scores'])
    assert len(out['boxes']) > 0
    torch.testing.assert_close(out['boxes'], torch.cat(want_b).clamp(0, 256) * 0.5)
    assert torch.equal(out['scores'], torch.cat(want_s))
    dup = evaluation.inference_on_slide(dep, slide, tile=128, overlap=64, batch_size=4)['det']
    s = dup['scores']
    assert (s[:-1] >= s[1:]).all()                                 # NMS order
    again = ops.nms(dup['boxes'], dup['scores'], m.headers['det'].nms_params['iou_thres'])
    assert len(again) == len(dup['boxes'])                         # idempotent: nothing left to suppress


@pytest.mark.parametrize('ml', [False, True])
def test_det_outputs_kernel_equals_the_tensor_expressions(ml):
    """hdy_det_outputs (hierarchical scores, best class / objectness fallback, labels, compaction over the batch) against the reference's
    per-image tensor expressions (yolo_head.py:335-345, :473-479) on a three-level class tree, with empty images and ties between classes"""
    from hd_yolo_amd import ops
    g = torch.Generator().manual_seed(9)
    B, max_det, nc = 7, 40, 6
    n_keep = torch.tensor([5, 0, 40, 1, 0, 17, 33], dtype=torch.int32)
    scores = torch.rand((B, max_det, 1 + nc), generator=g)
    scores[2, :, 3] = scores[2, :, 2]                                  # ties: the first maximum wins, as torch.max
    boxes = torch.rand((B, max_det, 4), generator=g) * 100
    # tree 0 -> {1 -> {2, 3}, 4, 5 -> {6}}: descendants in the order Detect.get_descendants records them (children before their parent)
    descendants = {1: [2, 3], 5: [6], 0: [1, 2, 3, 4, 5, 6]}
    pairs = torch.tensor([(c, k) for k, v in descendants.items() for c in v], dtype=torch.int32, device=DEV)
    res = {'scores': scores.clone().to(DEV), 'boxes': boxes.to(DEV), 'n_keep': n_keep.to(DEV)}
    conf = 0.2
    bx, sc, lb = ops.det_outputs(res, nc, conf, pairs, ml)
    off = 0
    for b in range(B):
        n = int(n_keep[b])
        x = scores[b, :n].clone()
        for k, v in descendants.items():
            x[:, v] *= x[:, k:k + 1]
        assert torch.equal(bx[off:off + n].cpu(), boxes[b, :n])
        assert torch.equal(res['scores'][b, :n].cpu(), x)              # in place on the padded rows, like the reference on its own rows
        if ml:
            assert torch.equal(sc[off:off + n].cpu(), x) and torch.equal(lb[off:off + n].cpu(), x > conf)
        elif n:
            cs, cl = x[:, 1:].max(1)
            assert torch.equal(sc[off:off + n].cpu(), torch.where(cs > conf, cs, x[:, 0]))
            assert torch.equal(lb[off:off + n].cpu(), torch.where(cs > conf, cl + 1, torch.full_like(cl, -100)))
        off += n
