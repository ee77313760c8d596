"""Detection metrics for the validation entry point (reference: metayolo/models/metrics.py:19-84 ap_per_class,
:86-110 compute_ap, :251-408 APMeter).  Host-side numpy on at most max_det rows per tile: not a kernel target
(SURVEY.md §2 row 5g); APMeter keeps the reference's matching rules and stats dictionary (pinned by tests/golden/apmeter.npz)."""
import numpy as np
import torch

from .utils_general import box_iou


def compute_ap(recall, precision):
    """Area under the precision envelope, 101-point interpolation (COCO)."""
    mrec = np.concatenate(([0.0], recall, [1.0]))
    mpre = np.concatenate(([1.0], precision, [0.0]))
    mpre = np.flip(np.maximum.accumulate(np.flip(mpre)))
    x = np.linspace(0, 1, 101)
    y = np.interp(x, mrec, mpre)
    return float(np.sum((y[1:] + y[:-1]) * np.diff(x)) / 2.0), mpre, mrec     # trapezoid rule


def ap_per_class(tp, conf, pred_cls, target_cls, eps=1e-16):
    """tp (n_pred, n_iou) bool, conf / pred_cls (n_pred,), target_cls (n_true,) -> p, r, ap (n_cls, n_iou), f1, classes."""
    order = np.argsort(-conf)
    tp, conf, pred_cls = tp[order], conf[order], pred_cls[order]
    classes, nt = np.unique(target_cls, return_counts=True)
    ap = np.zeros((len(classes), tp.shape[1]))
    p, r = np.zeros(len(classes)), np.zeros(len(classes))
    for ci, c in enumerate(classes):
        sel = pred_cls == c
        if sel.sum() == 0 or nt[ci] == 0:
            continue
        tpc, fpc = tp[sel].cumsum(0), (1 - tp[sel]).cumsum(0)
        recall, precision = tpc / (nt[ci] + eps), tpc / (tpc + fpc)
        r[ci], p[ci] = recall[-1, 0], precision[-1, 0]
        for j in range(tp.shape[1]):
            ap[ci, j] = compute_ap(recall[:, j], precision[:, j])[0]
    f1 = 2 * p * r / (p + r + eps)
    return p, r, ap, f1, classes.astype(int)


class ConfusionMatrix:
    """Detection confusion matrix with a background row / column (reference: metrics.py:114-169; val_nuclei.py:123 builds one per
    task).  matrix[predicted class, true class]; index nc = background (a missed label counts in row nc, a detection that matched
    nothing in column nc).  Matching: pairs with IoU > iou_thres; every detection keeps its best label, then every label its best
    detection."""

    def __init__(self, nc, conf=0.25, iou_thres=0.45):
        self.matrix = np.zeros((nc + 1, nc + 1))
        self.nc, self.conf, self.iou_thres = nc, conf, iou_thres

    def process_batch(self, detections, labels):
        """detections (n, 6) x1, y1, x2, y2, conf, class; labels (m, 5) class, x1, y1, x2, y2."""
        detections = detections[detections[:, 4] > self.conf]
        true_cls = labels[:, 0].int().tolist()
        det_cls = detections[:, 5].int().tolist()
        iou = box_iou(labels[:, 1:].float().cpu(), detections[:, :4].float().cpu()).numpy()       # (labels, detections)
        li, di = np.nonzero(iou > self.iou_thres)
        order = np.argsort(-iou[li, di], kind='stable')
        label_of = {}                                   # detection -> its best label
        for k in order:
            label_of.setdefault(int(di[k]), (int(li[k]), float(iou[li[k], di[k]])))
        det_of = {}                                     # label -> its best detection among those
        for d, (l, v) in sorted(label_of.items(), key=lambda kv: -kv[1][1]):
            det_of.setdefault(l, d)
        for l, c in enumerate(true_cls):
            if l in det_of:
                self.matrix[det_cls[det_of[l]], c] += 1
            else:
                self.matrix[self.nc, c] += 1
        if det_of:                                      # (with no match at all the reference counts no unmatched detections)
            used = set(det_of.values())
            for d, c in enumerate(det_cls):
                if d not in used:
                    self.matrix[c, self.nc] += 1

    def tp_fp(self):
        tp = self.matrix.diagonal()
        return tp[:-1], (self.matrix.sum(1) - tp)[:-1]


def summarize_precision_recall(stats_list, labels_text):
    """Pool per-image rows {label: (n_matched, n_true, n_pred, mean IoU)} into per-label precision / recall / F1 / mean IoU
    (reference: metrics.py:601-616)."""
    pooled = {}
    for stat in stats_list:
        for k, v in stat.items():
            pooled.setdefault(k, []).append(v)
    out = {}
    for k, rows in pooled.items():
        rows = np.array(rows)
        matched, n_true, n_pred = rows[:, 0].sum(), rows[:, 1].sum(), rows[:, 2].sum()
        precision = matched / n_pred if n_pred > 0 else np.nan
        recall = matched / n_true if n_true > 0 else np.nan
        out[labels_text[k]] = {'precision': precision, 'recall': recall, 'f1': 2 * precision * recall / (precision + recall),
                               'miou': rows[:, 3].mean()}
    return out


class APMeter:
    """Dataset-level detection AP with the reference's accumulation and matching rules (metayolo/models/metrics.py:251-375).

    add(): per image, predictions are put in descending score order; every (prediction, truth) pair with IoU >= 0.5 is recorded
    with dataset-global indices, the image's pairs in descending IoU order.
    ap_per_class(): pairs touching an ignored label are dropped; each prediction keeps its first (= best-IoU) pair, then each
    truth keeps the pair of its lowest-index (= best-score) prediction; a pair counts only when the two labels agree; a
    prediction is a true positive at threshold t when its pair's IoU >= t.  Predictions whose only pairs were with ignored
    truths are removed from the precision/recall curves.  Returns the reference's stats dict:
    'labels', 'counts', 'px', 'py' (n_cls, 1000), 'ap' (n_cls, n_iou), 'p', 'r', 'f1' (n_cls, 1000).
    Host-side numpy on <= max_det rows per tile (not a kernel target)."""

    def __init__(self, labels_text={}):
        self.iouv = np.linspace(0.5, 0.95, 10)
        self.labels_text = labels_text
        self.reset()

    def reset(self):
        self.n_pred = self.n_true = 0
        self._scores, self._y_pred, self._y_true = [], [], []
        self._m_pred, self._m_true, self._ious = [], [], []

    # the reference exposes these as tensors; keep the names readable from outside
    @property
    def scores(self):
        return np.concatenate(self._scores) if self._scores else np.zeros(0, np.float32)

    @property
    def y_pred(self):
        return np.concatenate(self._y_pred) if self._y_pred else np.zeros(0, np.int64)

    @property
    def y_true(self):
        return np.concatenate(self._y_true) if self._y_true else np.zeros(0, np.int64)

    @property
    def n_match(self):
        return int(sum(len(v) for v in self._ious))

    def add(self, output, target, iou_type='boxes'):
        if iou_type == 'masks' and 'masks' in output and 'masks' in target:
            raise NotImplementedError('mask IoU: the reference calls get_mask_ious here (metrics.py:275), a function its metrics module neither defines nor imports')
        scores, order = torch.sort(output['scores'].detach().float().cpu(), descending=True)
        boxes = output['boxes'].detach().float().cpu()[order]
        labels = output['labels'].detach().cpu()[order]
        tboxes, tlabels = target['boxes'].detach().float().cpu(), target['labels'].detach().cpu()
        iou = box_iou(boxes, tboxes).numpy() if len(boxes) and len(tboxes) else np.zeros((len(boxes), len(tboxes)), np.float32)
        pi, ti = np.nonzero(iou >= self.iouv.min())
        v = iou[pi, ti]
        o = np.argsort(-v, kind='stable')
        self._m_pred.append(pi[o] + self.n_pred)
        self._m_true.append(ti[o] + self.n_true)
        self._ious.append(v[o].astype(np.float32))
        self._y_true.append(tlabels.numpy().astype(np.int64))
        self._y_pred.append(labels.numpy().astype(np.int64))
        self._scores.append(scores.numpy())
        self.n_pred += len(boxes)
        self.n_true += len(tboxes)

    def ap_per_class(self, iouv=None, ignore=(-100, -1), eps=1e-16):
        # thresholds compare in fp32, as with the torch.linspace the reference's caller passes (val_nuclei.py:56)
        iouv = np.asarray(self.iouv if iouv is None else iouv, dtype=np.float32)
        cat = lambda parts, dt: np.concatenate(parts).astype(dt) if parts else np.zeros(0, dt)   # noqa: E731
        m_pred, m_true, ious = cat(self._m_pred, np.int64), cat(self._m_true, np.int64), cat(self._ious, np.float32)
        y_true, y_pred, scores = self.y_true, self.y_pred, self.scores
        ignore = list(ignore) if ignore else []

        ignored = (np.isin(y_true[m_true], ignore) | np.isin(y_pred[m_pred], ignore)) if ignore else np.zeros(len(m_pred), bool)
        k_pred, k_true, k_iou = m_pred[~ignored], m_true[~ignored], ious[~ignored]
        first = np.unique(k_pred, return_index=True)[1]            # one pair per prediction: its first = highest IoU
        k_pred, k_true, k_iou = k_pred[first], k_true[first], k_iou[first]
        first = np.unique(k_true, return_index=True)[1]            # one pair per truth: lowest prediction index
        k_pred, k_true, k_iou = k_pred[first], k_true[first], k_iou[first]
        agree = y_true[k_true] == y_pred[k_pred]
        k_pred, k_iou = k_pred[agree], k_iou[agree]
        hit = np.zeros((self.n_pred, len(iouv)), dtype=bool)
        hit[k_pred] = k_iou[:, None] >= iouv[None]

        if ignored.any():
            live = np.ones(self.n_pred, dtype=bool)
            live[np.setdiff1d(m_pred[ignored], k_pred)] = False
            hit, scores, y_pred = hit[live], scores[live], y_pred[live]
        order = np.argsort(-scores, kind='stable')
        hit, scores, y_pred = hit[order], scores[order], y_pred[order]

        px = np.linspace(0, 1, 1000)
        out = {'labels': [], 'counts': [], 'px': px}
        py, ap, p, r = [], [], [], []
        for c, n_true in zip(*np.unique(y_true, return_counts=True)):
            if c in ignore:
                continue
            out['labels'].append(c)
            out['counts'].append(n_true)
            sel = y_pred == c
            if sel.sum() == 0 or n_true == 0:
                ap.append(np.zeros(len(iouv)))
                for curve in (r, p, py):
                    curve.append(np.zeros(len(px)))
                continue
            tpc, fpc = hit[sel].cumsum(0), (~hit[sel]).cumsum(0)
            # fp32 curves, as the reference's torch arithmetic produces them: with fp64 a recall of exactly k/n can land ON a
            # knot of compute_ap's 101-point grid where the fp32 value falls just beside it, and the AP moves in the 4th digit
            tp32 = tpc.astype(np.float32)
            recall, precision = tp32 / np.float32(n_true + eps), tp32 / (tpc + fpc).astype(np.float32)
            r.append(np.interp(-px, -scores[sel], recall[:, 0], left=0))
            p.append(np.interp(-px, -scores[sel], precision[:, 0], left=1))
            row = np.zeros(len(iouv))
            for j in range(len(iouv)):
                row[j], mpre, mrec = compute_ap(recall[:, j], precision[:, j])
                if j == 0:
                    py.append(np.interp(px, mrec, mpre))
            ap.append(row)
        stack = lambda rows, w: np.stack(rows) if rows else np.zeros((0, w))   # noqa: E731
        out.update(py=stack(py, len(px)), ap=stack(ap, len(iouv)), p=stack(p, len(px)), r=stack(r, len(px)))
        out['f1'] = 2 * out['p'] * out['r'] / (out['p'] + out['r'] + eps)
        return out
