"""Detection loss with the reference's module names and buffer keys
(reference: metayolo/models/loss.py:20-48 smooth_label / WeightReduceLoss, :68-94 FocalLoss, :124-244 DetLoss).

SURVEY.md §8 row f1 ("next"): target assignment, CIoU and the BCE terms stay tensor expressions on the GPU for now;
they consume the fp32 logits the HIP plan produces and seed its backward.  One deliberate difference: the objectness
target scatter is made deterministic (last write wins, which is what the reference's CPU path does) instead of the
GPU's unordered index_put.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .utils_general import bbox_iou


def smooth_label(x, eps=0.1):
    return x - (x - 0.5) * eps


class WeightReduceLoss(nn.Module):
    """Element-wise loss_fn times a registered weight, then none / mean / sum."""

    def __init__(self, loss_fn, weight=None, reduction='mean'):
        super().__init__()
        self.loss_fn = loss_fn
        self.register_buffer('weight', weight)
        self.reduction = reduction

    def forward(self, input, target):
        loss = self.loss_fn(input, target)
        if self.weight is not None:
            loss = loss * self.weight
        if self.reduction == 'none':
            return loss
        return loss.mean() if self.reduction == 'mean' else loss.sum()


class FocalLoss(nn.Module):
    """Focal modulation around a BCE-with-logits criterion."""

    def __init__(self, loss_fcn, gamma=1.5, alpha=0.25):
        super().__init__()
        self.loss_fcn, self.gamma, self.alpha = loss_fcn, gamma, alpha
        self.reduction = loss_fcn.reduction
        self.loss_fcn.reduction = 'none'

    def forward(self, pred, true):
        loss = self.loss_fcn(pred, true)
        p = torch.sigmoid(pred)
        p_t = true * p + (1 - true) * (1 - p)
        loss = loss * (true * self.alpha + (1 - true) * (1 - self.alpha)) * (1.0 - p_t) ** self.gamma
        if self.reduction == 'mean':
            return loss.mean()
        return loss.sum() if self.reduction == 'sum' else loss


def scatter_last(dst, index_tuple, values):
    """dst[index_tuple] = values where, among duplicate indices, the LAST occurrence wins (sequential semantics)."""
    b, a, gj, gi = index_tuple
    shape = dst.shape
    lin = ((b * shape[1] + a) * shape[2] + gj) * shape[3] + gi
    order = torch.arange(lin.numel(), device=lin.device)
    last = torch.full((dst.numel(),), -1, dtype=torch.long, device=lin.device)
    last.scatter_reduce_(0, lin, order, reduce='amax', include_self=True)
    win = last[lin] == order
    dst.view(-1)[lin[win]] = values[win]
    return dst


class DetLoss(nn.Module):
    def __init__(self, nc, nl, hyp={}, ssi=0):
        super().__init__()
        self.gr, self.sort_obj_iou = 1.0, False
        self.nc, self.nl = nc, nl
        self.hyp = self.get_hyp_params(hyp)
        cls_weight = torch.tensor(hyp['cls_cw'])
        BCEcls = WeightReduceLoss(nn.BCEWithLogitsLoss(pos_weight=torch.tensor(hyp['cls_pw']), reduction='none'), cls_weight)
        BCEobj = nn.BCEWithLogitsLoss(pos_weight=torch.tensor(hyp['obj_pw']))
        if hyp['fl_gamma'] > 0.:
            BCEcls, BCEobj = FocalLoss(BCEcls, hyp['fl_gamma']), FocalLoss(BCEobj, hyp['fl_gamma'])
        self.BCEcls, self.BCEobj = BCEcls, BCEobj
        self.balance = {3: [4.0, 1.0, 0.4]}.get(self.nl, [4.0, 1.0, 0.25, 0.06, .02])
        self.autobalance, self.ssi = ssi > 0, ssi - 1

    def get_hyp_params(self, args={}):
        defaults = {'box': 0.05, 'cls': 0.05, 'obj': 1.0, 'cls_pw': 1.0, 'obj_pw': 1.0, 'cls_cw': 1.0, 'fl_gamma': 0.0,
                    'iou_t': 0.20, 'anchor_t': 4.0, 'label_smoothing': 0.0}
        return {k: args.get(k, v) for k, v in defaults.items()}

    def forward(self, p, tcls, tbox, indices, anchors):
        """p: per-level logits (bs, na, ny, nx, no); tcls / tbox / indices / anchors from Detect.matcher."""
        dev = p[0].device
        lcls, lbox, lobj = (torch.zeros(1, device=dev) for _ in range(3))
        for i, pi in enumerate(p):
            b, a, gj, gi = indices[i]
            tobj = torch.zeros(pi.shape[:4], dtype=pi.dtype, device=dev)
            if b.shape[0]:
                ps = pi[b, a, gj, gi]
                pxy = ps[:, 0:2].sigmoid() * 2 - 0.5
                pwh = (ps[:, 2:4].sigmoid() * 2) ** 2 * anchors[i]
                iou = bbox_iou(torch.cat((pxy, pwh), 1), tbox[i], CIoU=True).squeeze(-1)
                lbox = lbox + (1.0 - iou).mean()
                t = iou.detach().clamp(0).type(tobj.dtype)
                if self.sort_obj_iou:                   # loss.py:212-214: ascending IoU order, so the cell keeps its best match (last write wins)
                    j = t.argsort(stable=True)
                    b, a, gj, gi, t = b[j], a[j], gj[j], gi[j], t[j]
                if self.gr < 1:
                    t = (1.0 - self.gr) + self.gr * t
                scatter_last(tobj, (b, a, gj, gi), t)
                if self.nc > 1:
                    has = tcls[i][:, 1:].sum(-1) > 0
                    if has.any():
                        target = smooth_label(tcls[i][has][:, 1:], self.hyp['label_smoothing'])
                        lcls = lcls + self.BCEcls(ps[:, 5:][has], target).mean()
            obji = self.BCEobj(pi[..., 4], tobj)
            lobj = lobj + obji * self.balance[i]
            if self.autobalance:
                self.balance[i] = self.balance[i] * 0.9999 + 0.0001 / obji.detach().item()
        if self.autobalance:
            self.balance = [x / self.balance[self.ssi] for x in self.balance]
        lbox, lobj, lcls = lbox * self.hyp['box'], lobj * self.hyp['obj'], lcls * self.hyp['cls']
        bs = p[0].shape[0]
        return (lbox + lobj + lcls) * bs, {'box': lbox.detach(), 'obj': lobj.detach(), 'cls': lcls.detach()}


class SegLoss(nn.Module):
    """Mask loss on the (n, nc_masks, 28, 28) logits of the kept proposals (reference: metayolo/models/loss.py:247-283): the channel
    of each proposal's mask label, BCE-with-logits (or 1 - soft dice for type 'dice') against the 28 x 28 target masks, over proposals
    with a non-empty target and a label >= 0; times hyp['mask']; shape (1,) like det_loss."""

    def __init__(self, hyp={}):
        super().__init__()
        self.hyp = self.get_hyp_params(hyp)
        if self.hyp['type'] not in ('bce', 'dice'):
            raise ValueError(f"SegLoss type {self.hyp['type']!r}")

    def get_hyp_params(self, args={}):
        defaults = {'mask': 1.0, 'type': 'bce'}
        return {k: args.get(k, v) for k, v in defaults.items()}

    def forward(self, mask_logits, mask_targets, mask_labels):
        rows = torch.arange(mask_labels.shape[0], device=mask_labels.device)
        logits = mask_logits[rows, mask_labels][:, None]
        keep = (mask_targets.sum(dim=[1, 2, 3]) > 0) & (mask_labels >= 0)
        targets, logits = mask_targets[keep], logits[keep]
        if targets.numel() == 0:
            return logits.sum() * 0
        if self.hyp['type'] == 'bce':
            loss = F.binary_cross_entropy_with_logits(logits, targets)
        else:
            p = logits.sigmoid()                     # soft dice: mask_iou(factor=0, eps=0), utils_general.py:268-280
            prod, plus = (targets * p).sum([2, 3]), (targets + p).sum([2, 3])
            loss = 1 - (2 * prod / plus).mean()
        return (loss * self.hyp['mask'])[None]
