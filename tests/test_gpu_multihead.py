"""Several Detect headers on one backbone / neck (reference: metayolo/models/yolo.py:62-81 loops over self.headers; its hub files
name the scheme "multihead").  Pinned against single-header models that share the weights: per-task outputs and losses must be those
of the lone models, and the shared layers must receive the SUM of the two tasks' gradients."""
import os
import sys
from copy import deepcopy

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hd_yolo_amd import synth  # noqa: E402
from metayolo.models.yolo import Model  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda', 0)


def cfg_with(headers):
    cfg = synth.make_cfg('n', 2)
    base = cfg['headers'][0]
    cfg['headers'] = [[base[0], base[1], base[2], ['anchors', base[3][1], nc, -1], name] for name, nc in headers]
    return cfg


def build(headers):
    hyp = synth.make_hyp(conf_thres=0.05)
    hyp = {name: deepcopy(hyp['det']) for name, _ in headers}          # the hyp file holds one section per task tag (yolov5.py:105-110)
    m = Model(cfg_with(headers), hyp)
    m.load_state_dict(synth.synth_state_dict(synth.shapes_of(m), seed=0), strict=False)
    return m.to(DEV)


def relmax(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def targets_for(names_nc, B, S):
    per = {name: synth.synth_targets(B, S, nc, nmin=3, nmax=8, seed=5 + i) for i, (name, nc) in enumerate(names_nc)}
    out = []
    for b in range(B):
        anns = {}
        for name, _ in names_nc:
            anns[name] = [{k: v.to(DEV) for k, v in a.items()} for a in per[name][b]['anns']['det']]
        out.append({'anns': anns})
    return out


def test_two_headers_equal_two_single_header_models():
    heads = [('det', 2), ('aux', 3)]
    B, S = 2, 64
    multi = build(heads)
    singles = {name: build([(name, nc)]) for name, nc in heads}
    for name, m1 in singles.items():          # same weights everywhere: shared layers by construction, header layers copied over
        sd = {k: v for k, v in multi.state_dict().items() if not k.startswith('headers.') or k.startswith(f'headers.{name}.')}
        missing = m1.load_state_dict(sd, strict=False)
        assert not missing.unexpected_keys
    x = synth.synth_images(B, S, seed=7).to(DEV)

    # ---- inference: per-task detections are those of the lone model
    multi.eval()
    with torch.no_grad():
        _, out = multi(x)
    for name, m1 in singles.items():
        m1.eval()
        with torch.no_grad():
            _, o1 = m1(x)
        for a, b in zip(out, o1):
            for k in ('boxes', 'scores', 'labels'):
                assert torch.equal(a[name][k], b[name][k]), (name, k)

    # ---- training: per-task losses equal, shared-layer gradients add up
    targets = targets_for(heads, B, S)
    multi.train()
    losses, _ = multi(x, targets)
    sum(v['det_loss'] for v in losses.values()).backward()
    gsum = {}
    for name, m1 in singles.items():
        m1.train()
        l1, _ = m1(x, [{'anns': {name: t['anns'][name]}} for t in targets])
        a, b = float(l1[name]['det_loss'].detach()), float(losses[name]['det_loss'].detach())
        assert abs(a - b) <= 2e-4 * abs(a), (name, a, b)
        l1[name]['det_loss'].backward()
        for k, q in m1.named_parameters():
            if q.grad is not None:
                gsum[k] = gsum.get(k, 0) + q.grad.detach().clone()
    worst = 0.0
    for k, q in multi.named_parameters():
        assert q.grad is not None, k
        worst = max(worst, relmax(q.grad, gsum[k]))
    assert worst < 2e-3, worst
