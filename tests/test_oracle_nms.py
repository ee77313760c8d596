"""Known-answer tests for the NMS oracle (oracle/nms_ref.{py,c}).  torchvision is not available and
the reference has no NMS tests, so these hand-computed cases are what anchors "bit-exact NMS order"."""
import numpy as np
import pytest

from hd_yolo_amd import synth
from oracle import nms_ref


def both(boxes, scores, thr):
    a = nms_ref.nms_numpy(boxes, scores, thr)
    b = nms_ref.nms_c(boxes, scores, thr)
    assert np.array_equal(a, b)
    return a.tolist()


def test_docstring_example():
    # metayolo/models/utils_general.py:303-307: obj_1 p=0.9 box [0,0,50,50]; obj_2 p=0.8 box [1,1,51,51]
    # IoU = 49*49 / (2*2500 - 2401) = 0.9238 > 0.45 -> only obj_1 survives (ranked by objectness)
    preds = np.array([[[25, 25, 50, 50, 0.9, 0.5, 0.6, 0.0],
                       [26, 26, 50, 50, 0.8, 0.9, 0.1, 1.0]]], dtype=np.float32)
    out = nms_ref.nms_per_image_numpy(preds, nc=2, conf_thres=0.25, iou_thres=0.45)[0]
    assert out['index'].tolist() == [0]
    np.testing.assert_allclose(out['boxes'], [[0, 0, 50, 50]])
    np.testing.assert_allclose(out['scores'], [[0.9, 0.5, 0.6]])
    keep, nk, _ = nms_ref.nms_batched_c(preds, 2, 0.25, 0.45, 300)
    assert nk[0] == 1 and keep[0, 0] == 0
    # the class-aware variant ranks by obj*cls: obj_2/cls_1 = 0.72 beats obj_1/cls_2 = 0.54,
    # and the two boxes are in different classes so both are kept, best first
    keep, nk, cls = nms_ref.nms_batched_c(preds, 2, 0.25, 0.45, 300, class_aware=True)
    assert nk[0] == 2 and keep[0, :2].tolist() == [1, 0] and cls[0, :2].tolist() == [0, 1]


def test_chain_suppression_is_greedy_not_transitive():
    # A overlaps B (suppressed), B overlaps C, A does not overlap C -> C is kept
    boxes = np.array([[0, 0, 10, 10], [4, 0, 14, 10], [8, 0, 18, 10]], dtype=np.float32)
    scores = np.array([0.9, 0.8, 0.7], dtype=np.float32)
    # IoU(A,B) = 60/140 = 0.4286, IoU(B,C) same, IoU(A,C) = 20/180 = 0.111
    assert both(boxes, scores, 0.4) == [0, 2]
    assert both(boxes, scores, 0.45) == [0, 1, 2]


def test_threshold_is_strict():
    # IoU exactly 0.5: boxes [0,0,2,1] and [0,0,1,1] -> inter 1, union 2
    boxes = np.array([[0, 0, 2, 1], [0, 0, 1, 1]], dtype=np.float32)
    scores = np.array([0.9, 0.8], dtype=np.float32)
    assert both(boxes, scores, 0.5) == [0, 1]          # 0.5 > 0.5 is false: kept
    assert both(boxes, scores, 0.4999) == [0]


def test_ties_are_stable_descending():
    boxes = np.array([[0, 0, 10, 10], [100, 100, 110, 110], [0, 0, 10, 10], [200, 0, 210, 10]], dtype=np.float32)
    scores = np.array([0.5, 0.7, 0.5, 0.7], dtype=np.float32)
    # order: 1, 3 (ties by original index), then 0, 2 -> 2 suppressed by 0 (identical box)
    assert both(boxes, scores, 0.45) == [1, 3, 0]


def test_filters_small_and_low_conf_and_max_det():
    preds = np.array([[[10, 10, 1.9, 8, 0.9, 1, 0],      # w < 2 -> removed
                       [30, 30, 8, 8, 0.15, 1, 0],       # obj == conf -> removed (strict >)
                       [50, 50, 2.0, 2.0, 0.2, 1, 1],    # w == h == 2 -> kept
                       [70, 70, 8, 8, 0.3, 1, 2],
                       [90, 90, 8, 8, 0.4, 1, 0]]], dtype=np.float32)
    out = nms_ref.nms_per_image_numpy(preds, nc=1, conf_thres=0.15, iou_thres=0.45, max_det=2)[0]
    assert out['index'].tolist() == [4, 3]
    assert out['extra'].reshape(-1).tolist() == [0, 2]
    keep, nk, _ = nms_ref.nms_batched_c(preds, 1, 0.15, 0.45, 2)
    assert nk[0] == 2 and keep[0].tolist() == [4, 3]


def test_empty_inputs():
    assert both(np.zeros((0, 4), np.float32), np.zeros((0,), np.float32), 0.5) == []
    preds = np.zeros((2, 0, 8), dtype=np.float32)
    keep, nk, _ = nms_ref.nms_batched_c(preds, 2, 0.15, 0.45, 10)
    assert nk.tolist() == [0, 0]
    preds = np.zeros((1, 5, 8), dtype=np.float32)    # all below conf
    keep, nk, _ = nms_ref.nms_batched_c(preds, 2, 0.15, 0.45, 10)
    assert nk.tolist() == [0]


@pytest.mark.parametrize('m', [64, 500, 3000])
def test_numpy_and_c_agree_on_dense_tiles(m):
    preds = synth.synth_nms_preds(2, m, nc=8, extra=500).numpy()
    ref = nms_ref.nms_per_image_numpy(preds, 8, 0.15, 0.45, 300)
    keep, nk, _ = nms_ref.nms_batched_c(preds, 8, 0.15, 0.45, 300)
    for b in range(2):
        assert nk[b] == len(ref[b]['index'])
        assert np.array_equal(keep[b, :nk[b]], ref[b]['index'])
        assert 0 < nk[b] <= 300
    # sortedness + idempotence properties: kept scores non-increasing; NMS of the kept set keeps all
    for b in range(2):
        s = preds[b, keep[b, :nk[b]], 4]
        assert np.all(s[:-1] >= s[1:])
        again = nms_ref.nms_numpy(ref[b]['boxes'], s, 0.45)
        assert again.tolist() == list(range(len(s)))


def test_non_max_suppression_options_match_reference():
    """The oracle's statement-by-statement non_max_suppression against the reference's own function run with every option
    (tests/golden/nms_options.npz, make_golden.py:gen_nms_options; its greedy step there is this oracle's nms)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'nms_options.npz'))
    labels = [g['labels_0'], g['labels_1']]
    cases = {'default': {}, 'multi': {'multi_label': True}, 'agnostic': {'agnostic': True}, 'classes': {'classes': [0, 2]},
             'multi_agnostic_top5': {'multi_label': True, 'agnostic': True, 'max_det': 5}, 'apriori': {'labels': labels}}
    for tag, kw in cases.items():
        res = nms_ref.non_max_suppression_numpy(g['preds'], conf_thres=0.2, iou_thres=0.5, **kw)
        for b, d in enumerate(res):
            np.testing.assert_array_equal(d, g[f'{tag}_{b}'], err_msg=f'{tag} image {b}')
