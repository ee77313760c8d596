"""The CPU restatement of hnet's segmentation header (oracle/seg_ref.py) against the reference's own PanopticFeatureConnector /
PanopticSeg (tests/golden/seg.npz, written by tests/golden/make_golden.py seg)."""
import os
from collections import OrderedDict

import numpy as np
import torch

from oracle import seg_ref

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'seg.npz'))


def close(a, b, rtol=1e-5, atol=1e-6):
    np.testing.assert_allclose(a.detach().numpy() if torch.is_tensor(a) else a, b, rtol=rtol, atol=atol)


def test_connector_forward_and_gradients():
    names = ['23', '26', '29', '32']
    params = {k[len('con_p_'):]: torch.from_numpy(G[k]).requires_grad_(True) for k in G.files if k.startswith('con_p_')}
    feats = OrderedDict((n, torch.from_numpy(G[f'con_in_{n}']).requires_grad_(True)) for n in names)
    y = seg_ref.connector(params, feats)
    close(y, G['con_out'], atol=1e-5)
    (y * torch.from_numpy(G['con_wsum'])).sum().backward()
    for n in names:
        close(feats[n].grad, G[f'con_din_{n}'], rtol=1e-4, atol=1e-5)
    for k, v in params.items():
        close(v.grad, G[f'con_g_{k}'], rtol=1e-4, atol=2e-5)


def test_panoptic_loss_probabilities_and_gradients():
    names = ['17', '20', '23']
    params = {k[len('seg_p_'):]: torch.from_numpy(G[k]).requires_grad_(True) for k in G.files if k.startswith('seg_p_')}
    feats = OrderedDict((n, torch.from_numpy(G[f'seg_in_{n}']).requires_grad_(True)) for n in names)
    masks = torch.from_numpy(G['seg_masks'])
    probs, loss = seg_ref.panoptic(params, feats, 8, masks, class_weight=[1.0, 2.0, 0.5])
    close(loss.reshape(1), G['seg_loss'], rtol=1e-6)
    close(probs, G['seg_probs'], atol=1e-6)
    loss.backward()
    for n in names:
        close(feats[n].grad, G[f'seg_din_{n}'], rtol=1e-4, atol=1e-8)
    for k, v in params.items():
        close(v.grad, G[f'seg_g_{k}'], rtol=1e-4, atol=1e-7)
    with torch.no_grad():
        probs, _ = seg_ref.panoptic(params, feats, 8)
    close(probs, G['seg_eval_probs'], atol=1e-6)
    assert abs(float(probs.sum(1).mean()) - 1.0) < 1e-6
