"""Box utilities and NMS with the reference's names and argument meaning
(reference: metayolo/models/utils_general.py:12-28, :109-128, :161-191, :193-231, :247-296, :299-356, :423-523).

The two NMS entry points run on the MI355X kernel `hdy_nms_batched` (filter, stable sort, greedy suppression and
gather in one launch per batch); the small box-format helpers are plain tensor expressions used by host logic
(target preparation, metrics).  There is no CPU NMS here: CPU tensors raise.
Deliberate deviation: the reference's 10 s wall-clock bail-out (`utils_general.py:318,351-354`), which silently drops
the remaining images of a batch, is not reproduced.
"""
import math
from typing import Dict, List

import numpy as np
import torch

from .. import LOGGER
from ... import ops as _ops


# ---------------------------------------------------------------------------------------------- sizes
def make_divisible(x, divisor):
    if isinstance(divisor, torch.Tensor):
        divisor = int(divisor.max())
    return math.ceil(x / divisor) * divisor


def check_img_size(imgsz, s=32, floor=0):
    if isinstance(imgsz, int):
        new = max(make_divisible(imgsz, int(s)), floor)
    else:
        imgsz = list(imgsz)
        new = [max(make_divisible(v, int(s)), floor) for v in imgsz]
    if new != imgsz:
        LOGGER.warning(f'WARNING: --img-size {imgsz} must be multiple of max stride {s}, updating to {new}')
    return new


def labels_to_class_weights(labels, nc=80):
    """Inverse-frequency class weights, normalised so that present classes average 1 (reference :42-62)."""
    if labels[0] is None:
        return torch.Tensor()
    cls = np.concatenate(labels, 0)[:, 0].astype(int)
    counts = np.bincount(cls[cls >= 0], minlength=nc).astype(np.float64)
    present = counts > 0
    w = np.zeros_like(counts)
    w[present] = 1.0 / counts[present]
    return torch.from_numpy(w / w.sum() * present.sum())


def labels_to_image_weights(labels, nc=80, class_weights=np.ones(80)):
    per_image = []
    for lab in labels:
        cls = lab[:, 0].astype(int)
        per_image.append(np.bincount(cls[cls >= 0], minlength=nc))
    return (class_weights.reshape(1, nc) * np.array(per_image)).sum(1)


def coco80_to_coco91_class():
    skip = {12, 26, 29, 30, 45, 66, 68, 69, 71, 83}
    return [i for i in range(1, 91) if i not in skip]


# ---------------------------------------------------------------------------------------------- box formats
def _like(x):
    return x.clone() if isinstance(x, torch.Tensor) else np.copy(x)


def clip_coords(boxes, shape):
    """In-place clip of xyxy boxes to (height, width)."""
    if isinstance(boxes, torch.Tensor):
        boxes[:, 0].clamp_(0, shape[1])
        boxes[:, 1].clamp_(0, shape[0])
        boxes[:, 2].clamp_(0, shape[1])
        boxes[:, 3].clamp_(0, shape[0])
    else:
        boxes[:, [0, 2]] = boxes[:, [0, 2]].clip(0, shape[1])
        boxes[:, [1, 3]] = boxes[:, [1, 3]].clip(0, shape[0])


def xyxy2xywh(x, clip=False, eps=0.0):
    """corner -> centre format.  clip=True clamps the CALLER's tensor in place first, as the reference does."""
    if clip:
        clip_coords(x, (1.0 - eps, 1.0 - eps))
    y = _like(x)
    y[:, 0] = (x[:, 0] + x[:, 2]) / 2
    y[:, 1] = (x[:, 1] + x[:, 3]) / 2
    y[:, 2] = x[:, 2] - x[:, 0]
    y[:, 3] = x[:, 3] - x[:, 1]
    return y


def xywh2xyxy(x):
    y = _like(x)
    hw, hh = x[:, 2] / 2, x[:, 3] / 2
    y[:, 0], y[:, 1] = x[:, 0] - hw, x[:, 1] - hh
    y[:, 2], y[:, 3] = x[:, 0] + hw, x[:, 1] + hh
    return y


def xywhn2xyxy(x, w=640, h=640, padw=0, padh=0):
    y = _like(x)
    y[:, 0] = w * (x[:, 0] - x[:, 2] / 2) + padw
    y[:, 1] = h * (x[:, 1] - x[:, 3] / 2) + padh
    y[:, 2] = w * (x[:, 0] + x[:, 2] / 2) + padw
    y[:, 3] = h * (x[:, 1] + x[:, 3] / 2) + padh
    return y


def xyxy2xywhn(x, w=640, h=640, clip=False, eps=0.0):
    if clip:
        clip_coords(x, (h - eps, w - eps))
    y = _like(x)
    y[:, 0] = ((x[:, 0] + x[:, 2]) / 2) / w
    y[:, 1] = ((x[:, 1] + x[:, 3]) / 2) / h
    y[:, 2] = (x[:, 2] - x[:, 0]) / w
    y[:, 3] = (x[:, 3] - x[:, 1]) / h
    return y


def scale_coords(img1_shape, coords, img0_shape, ratio_pad=None):
    """Map xyxy boxes from the network input frame back to the original image frame (in place)."""
    if isinstance(img1_shape, int):
        img1_shape = (img1_shape, img1_shape)
    if isinstance(img0_shape, int):
        img0_shape = (img0_shape, img0_shape)
    if ratio_pad is None:
        gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
        pad = (img1_shape[1] - img0_shape[1] * gain) / 2, (img1_shape[0] - img0_shape[0] * gain) / 2
    else:
        gain, pad = ratio_pad[0][0], ratio_pad[1]
    coords[:, [0, 2]] -= pad[0]
    coords[:, [1, 3]] -= pad[1]
    coords[:, :4] /= gain
    clip_coords(coords, img0_shape)
    return coords


# ---------------------------------------------------------------------------------------------- IoU family
def bbox_iou(box1, box2, xywh=True, GIoU=False, DIoU=False, CIoU=False, eps=1e-7):
    """IoU / GIoU / DIoU / CIoU of (n,4) rows against (n,4) rows, result (n,1)."""
    if xywh:
        (x1, y1, w1, h1), (x2, y2, w2, h2) = box1.chunk(4, 1), box2.chunk(4, 1)
        a_x1, a_x2, a_y1, a_y2 = x1 - w1 / 2, x1 + w1 / 2, y1 - h1 / 2, y1 + h1 / 2
        b_x1, b_x2, b_y1, b_y2 = x2 - w2 / 2, x2 + w2 / 2, y2 - h2 / 2, y2 + h2 / 2
    else:
        a_x1, a_y1, a_x2, a_y2 = box1.chunk(4, 1)
        b_x1, b_y1, b_x2, b_y2 = box2.chunk(4, 1)
        w1, h1 = a_x2 - a_x1, a_y2 - a_y1 + eps
        w2, h2 = b_x2 - b_x1, b_y2 - b_y1 + eps
    inter = (torch.min(a_x2, b_x2) - torch.max(a_x1, b_x1)).clamp(0) * (torch.min(a_y2, b_y2) - torch.max(a_y1, b_y1)).clamp(0)
    union = w1 * h1 + w2 * h2 - inter + eps
    iou = inter / union
    if not (CIoU or DIoU or GIoU):
        return iou
    cw = torch.max(a_x2, b_x2) - torch.min(a_x1, b_x1)
    ch = torch.max(a_y2, b_y2) - torch.min(a_y1, b_y1)
    if GIoU and not (CIoU or DIoU):
        c_area = cw * ch + eps
        return iou - (c_area - union) / c_area
    c2 = cw ** 2 + ch ** 2 + eps
    rho2 = ((b_x1 + b_x2 - a_x1 - a_x2) ** 2 + (b_y1 + b_y2 - a_y1 - a_y2) ** 2) / 4
    if not CIoU:
        return iou - rho2 / c2
    v = (4 / math.pi ** 2) * torch.pow(torch.atan(w2 / h2) - torch.atan(w1 / h1), 2)
    with torch.no_grad():
        alpha = v / (v - iou + (1 + eps))
    return iou - (rho2 / c2 + v * alpha)


def box_area(box):
    return (box[2] - box[0]) * (box[3] - box[1])


def box_iou(box1, box2):
    """(N,4) x (M,4) xyxy -> (N,M) IoU matrix."""
    (a1, a2), (b1, b2) = box1[:, None].chunk(2, 2), box2.chunk(2, 1)
    inter = (torch.min(a2, b2) - torch.max(a1, b1)).clamp(0).prod(2)
    return inter / (box_area(box1.T)[:, None] + box_area(box2.T) - inter)


def wh_iou(wh1, wh2):
    """(n, m) IoU of boxes given by width / height only, as if they shared a corner (reference: utils_general.py:234-239)."""
    a, b = wh1[:, None, :], wh2[None, :, :]
    overlap = torch.minimum(a, b).prod(-1)
    return overlap / (a.prod(-1) + b.prod(-1) - overlap)


def mask_iou(y_pred, y_true, factor=0.0, axis=(2, 3), eps=0.):
    """Soft overlap of two mask stacks: (2 + factor) * sum(t*p) / (sum(t + p) + factor * sum(t*p) + eps); factor 0 (or 'dice') is
    the Dice coefficient, -1 (or 'iou') the IoU (reference: utils_general.py:268-280; SegLoss type 'dice' uses factor 0)."""
    factor = {'dice': 0.0, 'iou': -1.0}.get(factor, factor) if isinstance(factor, str) else factor
    both = (y_true * y_pred).sum(list(axis))
    total = (y_true + y_pred).sum(list(axis))
    return (2 + factor) * both / (total + factor * both + eps)


def xyn2xy(x, w=640, h=640, padw=0, padh=0):
    """Normalised (n, 2) points -> pixels, with an offset (reference: utils_general.py:153-158)."""
    y = x.clone() if isinstance(x, torch.Tensor) else np.copy(x)
    y[:, 0] = x[:, 0] * w + padw
    y[:, 1] = x[:, 1] * h + padh
    return y


def check_anchor_order(m):
    """Stock-YOLOv5 Detect layout (m.anchors (nl, na, 2), m.stride (nl,)): flip the anchor rows when their mean area runs against the
    stride order (reference: utils_general.py:31-38; metayolo's own Detect keeps per-level buffers and does not call it)."""
    area = m.anchors.prod(-1).mean(-1).view(-1)
    da, ds = area[-1] - area[0], m.stride[-1] - m.stride[0]
    if da and da.sign() != ds.sign():
        m.anchors[:] = m.anchors.flip(0)


def paired_box_iou(boxes1, boxes2):
    """Row-wise IoU of two (N,4) xyxy sets."""
    wh = (torch.min(boxes1[:, 2:], boxes2[:, 2:]) - torch.max(boxes1[:, :2], boxes2[:, :2])).clamp(min=0)
    inter = wh[:, 0] * wh[:, 1]
    a1 = (boxes1[:, 2] - boxes1[:, 0]) * (boxes1[:, 3] - boxes1[:, 1])
    a2 = (boxes2[:, 2] - boxes2[:, 0]) * (boxes2[:, 3] - boxes2[:, 1])
    return inter / (a1 + a2 - inter)


# ---------------------------------------------------------------------------------------------- NMS
def nms(boxes, scores, iou_threshold):
    """What the reference takes from torchvision.ops.nms (yolo.py:195): kept indices by descending score, IoU > threshold
    suppressed, class-agnostic.  Runs on the MI355X (hdy_nms_boxes); at most 4096 boxes are kept."""
    return _ops.nms(boxes, scores, iou_threshold)


def _check_thresholds(conf_thres, iou_thres):
    assert 0 <= conf_thres <= 1, f'Invalid Confidence threshold {conf_thres}, valid values are between 0.0 and 1.0'
    assert 0 <= iou_thres <= 1, f'Invalid IoU {iou_thres}, valid values are between 0.0 and 1.0'


def nms_per_image(preds: torch.Tensor, nc: int, conf_thres: float = 0.25, iou_thres: float = 0.45,
                  max_det: int = 300) -> List[Dict[str, torch.Tensor]]:
    """Class-agnostic NMS ranked by objectness, one result dict per image:
        {'boxes': (n,4) xyxy px, 'scores': (n,1+nc) [obj, cls...], 'extra': (n,E) trailing columns, 'index': (n,) kept rows}
    preds: (B, N, 5+nc+E) [cx, cy, w, h, obj, cls..., extra...].  Boxes with w or h < 2 px and rows with obj <= conf_thres are
    dropped first; suppression is IoU > iou_thres; kept rows come in descending obj order (ties: lower row first)."""
    _check_thresholds(conf_thres, iou_thres)
    _ops.require_gpu(preds)
    B = preds.shape[0]
    if B == 0:
        return []
    res = _ops.nms_batched(preds.float().contiguous(), nc, conf_thres, iou_thres, int(max_det), min_wh=2.0, class_aware=False)
    n_keep = res['n_keep'].tolist()                     # one D2H sync per batch
    out = []
    for b, n in enumerate(n_keep):
        out.append({'boxes': res['boxes'][b, :n], 'scores': res['scores'][b, :n], 'extra': res['extra'][b, :n],
                    'index': res['keep'][b, :n]})
    return out


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False,
                        labels=(), max_det=300):
    """Class-aware NMS on (B, N, 5+nc) predictions -> list of (n,6) [xyxy, conf, cls] per image (reference :423-523):
    conf = obj*cls, best class per box (or one candidate per (box, class) above the threshold with multi_label), boxes of
    different classes never suppress each other (class offset 7680 px) unless `agnostic`.
    The default options run as ONE batched launch of the NMS kernel (filter + sort + greedy + gather); the rarely used ones
    (multi_label, agnostic, classes, apriori labels) build the candidate list with tensor expressions per image and run the same
    kernel on explicit boxes (hdy_nms_boxes)."""
    _check_thresholds(conf_thres, iou_thres)
    _ops.require_gpu(prediction)
    nc = prediction.shape[2] - 5
    multi_label = bool(multi_label) and nc > 1
    has_labels = bool(len(labels)) and any(len(lb) for lb in labels)
    if not (multi_label or agnostic or has_labels or classes is not None) and max_det <= 4096:
        res = _ops.nms_batched(prediction.float().contiguous(), nc, conf_thres, iou_thres, int(max_det), class_aware=True)
        n_keep = res['n_keep'].tolist()
        return [torch.cat([res['boxes'][b, :n], res['conf'][b, :n, None], res['cls'][b, :n, None].float()], 1)
                for b, n in enumerate(n_keep)]
    out = []
    for xi, x in enumerate(prediction.float()):
        x = x[x[:, 4] > conf_thres]
        if has_labels and len(labels[xi]):
            lb = labels[xi].to(x.device, torch.float32)
            v = torch.zeros((len(lb), nc + 5), device=x.device)
            v[:, :4], v[:, 4] = lb[:, 1:5], 1.0
            v[torch.arange(len(lb)), lb[:, 0].long() + 5] = 1.0
            x = torch.cat((x, v), 0)
        det = torch.zeros((0, 6), device=prediction.device)
        if x.shape[0]:
            x = x.clone()
            x[:, 5:] *= x[:, 4:5]
            box = xywh2xyxy(x[:, :4])
            if multi_label:
                i, j = (x[:, 5:] > conf_thres).nonzero(as_tuple=False).T
                det = torch.cat((box[i], x[i, j + 5, None], j[:, None].float()), 1)
            else:
                conf, j = x[:, 5:].max(1, keepdim=True)
                det = torch.cat((box, conf, j.float()), 1)[conf.view(-1) > conf_thres]
            if classes is not None:
                det = det[(det[:, 5:6] == torch.tensor(classes, device=det.device)).any(1)]
        if det.shape[0] > 30000:
            det = det[det[:, 4].argsort(descending=True, stable=True)[:30000]]
        if det.shape[0]:
            offs = det[:, 5:6] * (0.0 if agnostic else 7680.0)
            keep = _ops.nms(det[:, :4] + offs, det[:, 4], iou_thres, max_det=int(max_det))
            det = det[keep[:max_det]]
        out.append(det)
    return out
