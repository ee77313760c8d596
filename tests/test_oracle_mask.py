"""The oracle's restatement of the mask branch's third-party pieces (oracle/mask_ref.py) against scalar known-answer code and the
mask goldens' own structure.  torchvision / torch_scatter are not under /root/reference: parity at this boundary is unpinned."""
import os

import numpy as np
import torch

from oracle import mask_ref


def _naive_roi_align(f, roi, P, scale, S, aligned):
    b, off = int(roi[0]), (0.5 if aligned else 0.0)
    x0, y0 = roi[1] * scale - off, roi[2] * scale - off
    rw, rh = roi[3] * scale - off - x0, roi[4] * scale - off - y0
    if not aligned:
        rw, rh = max(rw, 1.0), max(rh, 1.0)
    C, H, W = f.shape[1:]
    out = torch.zeros(C, P, P)
    for ph in range(P):
        for pw in range(P):
            acc = torch.zeros(C)
            for iy in range(S):
                y = y0 + ph * rh / P + (iy + 0.5) * rh / P / S
                for ix in range(S):
                    x = x0 + pw * rw / P + (ix + 0.5) * rw / P / S
                    if y < -1 or y > H or x < -1 or x > W:
                        continue
                    yy, xx = max(y, 0.0), max(x, 0.0)
                    yl, xl = int(yy), int(xx)
                    if yl >= H - 1:
                        yl = yh = H - 1
                        yy = float(yl)
                    else:
                        yh = yl + 1
                    if xl >= W - 1:
                        xl = xh = W - 1
                        xx = float(xl)
                    else:
                        xh = xl + 1
                    ly, lx = yy - yl, xx - xl
                    acc += (1 - ly) * (1 - lx) * f[b, :, yl, xl] + (1 - ly) * lx * f[b, :, yl, xh] + ly * (1 - lx) * f[b, :, yh, xl] \
                        + ly * lx * f[b, :, yh, xh]
            out[:, ph, pw] = acc / (S * S)
    return out


def test_roi_align_known_answers():
    torch.manual_seed(0)
    f = torch.randn(2, 3, 9, 11)
    rois = torch.tensor([[1, 2.3, 1.7, 30.2, 25.9], [0, -5.0, -3.0, 7.0, 4.0], [1, 60.0, 50.0, 90.0, 80.0], [0, 8.0, 8.0, 8.1, 8.2]])
    for aligned in (False, True):
        o = mask_ref.roi_align(f, rois, 4, 0.25, 2, aligned)
        for r in range(len(rois)):
            assert (o[r] - _naive_roi_align(f, rois[r].tolist(), 4, 0.25, 2, aligned)).abs().max() < 1e-5
    # a constant map pools to that constant wherever the samples fall inside; list-of-boxes form == (R, 5) form
    c = torch.full((1, 2, 8, 8), 3.0)
    assert torch.allclose(mask_ref.roi_align(c, torch.tensor([[0, 1.0, 1.0, 5.0, 6.0]]), 7, 1.0, 2), torch.full((1, 2, 7, 7), 3.0))
    a = mask_ref.roi_align(f, [rois[1:2, 1:], rois[0:1, 1:]], 4, 0.25, 2)
    assert torch.equal(a, mask_ref.roi_align(f, torch.cat([rois[1:2], rois[0:1]]), 4, 0.25, 2))
    assert mask_ref.roi_align(f, torch.zeros((0, 5)), 4, 0.25, 2).shape == (0, 3, 4, 4)


def test_scatter_max_and_head_shapes():
    v, i = mask_ref.scatter_max(torch.tensor([0.2, 0.9, 0.9, 0.1, 0.5]), torch.tensor([0, 2, 2, 0, 3]))
    assert v.tolist() == [0.20000000298023224, 0.0, 0.8999999761581421, 0.5] and i.tolist() == [0, 5, 1, 4]
    heads = mask_ref.MaskRCNNHeads(16, (256, 256, 256, 256), 1)
    pred = mask_ref.MaskRCNNPredictor(256, 256, 3)
    assert list(heads.state_dict())[:2] == ['mask_fcn1.weight', 'mask_fcn1.bias'] and 'mask_fcn_logits.bias' in pred.state_dict()
    assert pred(heads(torch.zeros(2, 16, 14, 14))).shape == (2, 3, 28, 28)


def test_mask_goldens_are_not_degenerate():
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'masks.npz'))
    assert float(g['train_mask_loss']) > 0.1
    names = g['gradsum_names'].tolist()
    assert g['gradsum'][names.index('headers.det.seg.2.conv.weight'), 2] > 0 and g['eval_mask_probs'].std() > 0.01
