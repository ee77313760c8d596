"""TEST INFRASTRUCTURE ONLY — CPU oracle for filter + greedy NMS.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Two independent restatements of the same algorithm, checked against each other and against
hand-computed known-answer cases in tests/test_oracle_nms.py:
  * nms_numpy / nms_per_image_numpy  — numpy, fp32, one vector op per visited box
  * oracle/nms_ref.c via ctypes        — scalar C (also the timed CPU baseline for NMS)

Reference call sites followed: metayolo/models/utils_general.py:299-356 (nms_per_image),
:423-523 (non_max_suppression), :121-128 (xywh2xyxy).  The greedy step is torchvision.ops.nms,
absent from /root/reference and un-pinned; see the header of nms_ref.c for the published
algorithm restated and the tie rule (stable descending).  PARITY: unpinned at that boundary.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, '_build')
_LIB = None


def build(force=False):
    """gcc the C restatement into oracle/_build/libhdy_oracle.so (no GPU, no torch)."""
    os.makedirs(_BUILD, exist_ok=True)
    so = os.path.join(_BUILD, 'libhdy_oracle.so')
    src = os.path.join(_HERE, 'nms_ref.c')
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['gcc', '-O2', '-ffp-contract=off', '-fno-fast-math', '-shared', '-fPIC',
                               '-o', so, src])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        L.hdy_ref_nms.restype = ctypes.c_int
        L.hdy_ref_nms.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int,
                                  ctypes.c_void_p]
        L.hdy_ref_nms_batched.restype = None
        L.hdy_ref_nms_batched.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_float,
                                          ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        _LIB = L
    return _LIB


def nms_numpy(boxes, scores, iou_thr):
    """torchvision.ops.nms semantics on xyxy fp32 boxes; returns int64 kept indices, score-descending."""
    boxes = np.ascontiguousarray(boxes, dtype=np.float32)
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    n = len(boxes)
    if n == 0:
        return np.zeros((0,), dtype=np.int64)
    x1, y1, x2, y2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    area = (x2 - x1) * (y2 - y1)
    order = np.argsort(-scores, kind='stable')
    sup = np.zeros(n, dtype=bool)
    thr = np.float32(iou_thr)
    keep = []
    for a in range(n):
        i = order[a]
        if sup[i]:
            continue
        keep.append(i)
        rest = order[a + 1:]
        w = np.maximum(np.float32(0), np.minimum(x2[i], x2[rest]) - np.maximum(x1[i], x1[rest]))
        h = np.maximum(np.float32(0), np.minimum(y2[i], y2[rest]) - np.maximum(y1[i], y1[rest]))
        inter = w * h
        with np.errstate(divide='ignore', invalid='ignore'):
            ovr = inter / (area[i] + area[rest] - inter)
        sup[rest[ovr > thr]] = True
    return np.asarray(keep, dtype=np.int64)


def xywh2xyxy_np(x):
    x = np.asarray(x, dtype=np.float32)
    hw, hh = x[:, 2] / np.float32(2), x[:, 3] / np.float32(2)
    return np.stack([x[:, 0] - hw, x[:, 1] - hh, x[:, 0] + hw, x[:, 1] + hh], 1)


def nms_per_image_numpy(preds, nc, conf_thres=0.25, iou_thres=0.45, max_det=300, min_wh=2.0):
    """utils_general.py:299-356 on a (B, N, 5+nc+extra) array -> list of dicts with the
    kept ORIGINAL row indices too ('index')."""
    out = []
    for x in np.asarray(preds, dtype=np.float32):
        boxes = xywh2xyxy_np(x[:, :4])
        idx = np.arange(len(x))
        k = ((boxes[:, 2] - boxes[:, 0]) >= np.float32(min_wh)) & ((boxes[:, 3] - boxes[:, 1]) >= np.float32(min_wh))
        k &= x[:, 4] > np.float32(conf_thres)
        boxes, rows, idx = boxes[k], x[k], idx[k]
        keep = nms_numpy(boxes, rows[:, 4], iou_thres)[:max_det]
        out.append({'boxes': boxes[keep], 'scores': rows[keep, 4:5 + nc], 'extra': rows[keep, 5 + nc:],
                    'index': idx[keep]})
    return out


def non_max_suppression_numpy(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False,
                              labels=(), max_det=300):
    """utils_general.py:423-523 with every option, on a (B, N, 5+nc) fp32 array -> list of (n, 6) [xyxy, conf, cls].
    Follows the reference statement by statement (candidate filter obj > conf, apriori label rows, conf = obj*cls, best class
    or multi-label expansion, class filter, 30000 pre-cut, class offset 7680 unless agnostic, greedy NMS, max_det)."""
    prediction = np.asarray(prediction, dtype=np.float32)
    nc = prediction.shape[2] - 5
    multi_label = bool(multi_label) and nc > 1
    out = []
    for xi, x in enumerate(prediction):
        x = x[x[:, 4] > np.float32(conf_thres)].copy()
        if len(labels) and len(labels[xi]):
            lb = np.asarray(labels[xi], dtype=np.float32)
            v = np.zeros((len(lb), nc + 5), dtype=np.float32)
            v[:, :4] = lb[:, 1:5]
            v[:, 4] = 1.0
            v[np.arange(len(lb)), lb[:, 0].astype(np.int64) + 5] = 1.0
            x = np.concatenate([x, v], 0)
        if not len(x):
            out.append(np.zeros((0, 6), dtype=np.float32))
            continue
        x[:, 5:] *= x[:, 4:5]
        box = xywh2xyxy_np(x[:, :4])
        if multi_label:
            i, j = np.nonzero(x[:, 5:] > np.float32(conf_thres))
            det = np.concatenate([box[i], x[i, j + 5, None], j[:, None].astype(np.float32)], 1)
        else:
            j = x[:, 5:].argmax(1)
            conf = x[np.arange(len(x)), j + 5]
            det = np.concatenate([box, conf[:, None], j[:, None].astype(np.float32)], 1)[conf > np.float32(conf_thres)]
        if classes is not None:
            det = det[np.isin(det[:, 5], np.asarray(classes, dtype=np.float32))]
        if not len(det):
            out.append(np.zeros((0, 6), dtype=np.float32))
            continue
        if len(det) > 30000:
            det = det[np.argsort(-det[:, 4], kind='stable')[:30000]]
        c = det[:, 5:6] * np.float32(0 if agnostic else 7680)
        keep = nms_numpy(det[:, :4] + c, det[:, 4], iou_thres)[:max_det]
        out.append(det[keep])
    return out


def nms_batched_c(preds, nc, conf_thres, iou_thres, max_det, min_wh=2.0, class_aware=False):
    """C restatement on a (B, N, row) fp32 array -> (keep [B,max_det] int64, n_keep [B] int32, cls [B,max_det])."""
    preds = np.ascontiguousarray(preds, dtype=np.float32)
    B, N, row = preds.shape
    keep = np.full((B, max_det), -1, dtype=np.int64)
    n_keep = np.zeros((B,), dtype=np.int32)
    cls = np.zeros((B, max_det), dtype=np.int32)
    lib().hdy_ref_nms_batched(preds.ctypes.data, B, N, row, nc, conf_thres, iou_thres, max_det, min_wh,
                              1 if class_aware else 0, keep.ctypes.data, n_keep.ctypes.data, cls.ctypes.data)
    return keep, n_keep, cls


def nms_c(boxes, scores, iou_thr, max_keep=None):
    boxes = np.ascontiguousarray(boxes, dtype=np.float32)
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    m = len(boxes)
    max_keep = m if max_keep is None else max_keep
    keep = np.zeros((max(max_keep, 1),), dtype=np.int32)
    nk = lib().hdy_ref_nms(boxes.ctypes.data, scores.ctypes.data, m, iou_thr, max_keep, keep.ctypes.data)
    return keep[:nk].astype(np.int64)
