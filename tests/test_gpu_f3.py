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
    assert len(want) <= 4096, 'test sized for the kept-list capacity'
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
