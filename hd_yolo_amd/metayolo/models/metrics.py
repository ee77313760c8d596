"""Detection metrics for the validation entry point (reference: metayolo/models/metrics.py:19-84 ap_per_class,
:86-110 compute_ap, :251-408 APMeter).  Host-side numpy on at most max_det rows per tile: not a kernel target
(SURVEY.md §2 row 5g); re-authored compactly with the same call surface (APMeter.add / ap_per_class)."""
import numpy as np
import torch

from .utils_general import box_iou


def compute_ap(recall, precision):
    """Area under the precision envelope, 101-point interpolation (COCO)."""
    mrec = np.concatenate(([0.0], recall, [1.0]))
    mpre = np.concatenate(([1.0], precision, [0.0]))
    mpre = np.flip(np.maximum.accumulate(np.flip(mpre)))
    x = np.linspace(0, 1, 101)
    return np.trapz(np.interp(x, mrec, mpre), x), mpre, mrec


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


class APMeter:
    """Accumulates (detections, ground truth) per image; one-to-one greedy matching by IoU, class-aware AP."""

    def __init__(self, labels_text={}):
        self.iouv = np.linspace(0.5, 0.95, 10)
        self.labels_text = labels_text
        self.reset()

    def reset(self):
        self.tp, self.scores, self.y_pred, self.y_true = [], [], [], []

    def add(self, output, target, iou_type='boxes'):
        scores, order = torch.sort(output['scores'].float(), descending=True)
        boxes, labels = output['boxes'][order].float(), output['labels'][order]
        tboxes, tlabels = target['boxes'].float().to(boxes.device), target['labels'].to(boxes.device)
        n_pred, n_true = boxes.shape[0], tboxes.shape[0]
        tp = np.zeros((n_pred, len(self.iouv)), dtype=bool)
        if n_pred and n_true:
            iou = box_iou(boxes, tboxes)
            same = labels[:, None] == tlabels[None]
            iou_np = (iou * same).cpu().numpy()
            for j, thr in enumerate(self.iouv):
                pi, ti = np.where(iou_np >= thr)
                if len(pi):
                    m = np.stack([pi, ti, iou_np[pi, ti]], 1)
                    m = m[np.argsort(-m[:, 2])]
                    m = m[np.unique(m[:, 0], return_index=True)[1]]
                    m = m[np.argsort(-m[:, 2])]
                    m = m[np.unique(m[:, 1], return_index=True)[1]]
                    tp[m[:, 0].astype(int), j] = True
        self.tp.append(tp)
        self.scores.append(scores.cpu().numpy())
        self.y_pred.append(labels.cpu().numpy())
        self.y_true.append(tlabels.cpu().numpy())

    def ap_per_class(self, iouv=None, ignore=(-100, -1)):
        tp = np.concatenate(self.tp) if self.tp else np.zeros((0, len(self.iouv)), dtype=bool)
        conf = np.concatenate(self.scores) if self.scores else np.zeros(0)
        pc = np.concatenate(self.y_pred) if self.y_pred else np.zeros(0)
        tc = np.concatenate(self.y_true) if self.y_true else np.zeros(0)
        keep_p, keep_t = ~np.isin(pc, ignore), ~np.isin(tc, ignore)
        p, r, ap, f1, classes = ap_per_class(tp[keep_p], conf[keep_p], pc[keep_p], tc[keep_t])
        return {'p': p, 'r': r, 'ap': ap, 'f1': f1, 'classes': classes, 'nt': np.bincount(tc[keep_t].astype(int))[classes] if len(classes) else np.zeros(0)}
