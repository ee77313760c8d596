"""TEST INFRASTRUCTURE ONLY — CPU (torch fp32) restatement of the metayolo detection hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product package (hd_yolo_amd/) never does and fails loudly without its HIP library.

It is a *functional* interpreter: weights live in a plain {state_dict key: tensor} mapping
with the reference's key names, and every op is an explicit torch.nn.functional call, so the
same weights can be fed to the HIP path and to this oracle.  PARITY: pinned — checked against
golden vectors produced by the reference's own code (tests/golden/make_golden.py,
tests/test_oracle_golden.py) for stage outputs, det logits, decode, outputs, loss and gradients.
Kept-box ORDER goes through oracle/nms_ref.py, which is unpinned at the torchvision boundary.

Reference lines restated (all under /root/reference/metayolo/models/):
  layers.py:18-41      Conv = SiLU(BN(conv2d)), autopad k//2, bias-free conv
  layers.py:87-97      Bottleneck = x + cv2(cv1(x)) when shortcut and c1 == c2
  layers.py:119-131    C3 = cv3(cat(m(cv1(x)), cv2(x)))
  layers.py:174-189    SPPF = cv2(cat(x, p(x), p(p(x)), p(p(p(x))))), p = maxpool 5/1/2
  layers.py:264-271    Concat
  yolov5.py:47-77      CSPDarkNet / FPN routing by `from` index
  yolov5.py:80-161     build_network: depth gain, width gain make_divisible(c*gw, 8)
  utils_torch.py:42-51 BN eps = 1e-3, momentum = 0.03
  utils_torch.py:79-99 fuse_conv_and_bn
  yolo_head.py:132-213 Detect det conv + reshape + decode;  :301-355 compute_outputs
  yolo_head.py:358-417 matcher;  :431-438 bias init;  :473-479 hierarchical scores
  loss.py:190-244      DetLoss.forward;  utils_general.py:193-231 bbox_iou(CIoU)
  utils_general.py:109-128 xyxy2xywh(clip) / xywh2xyxy
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import nms_ref

BN_EPS, BN_MOM = 1e-3, 0.03


def _ceil_to(x, d):
    return int(math.ceil(x / d) * d)


class RefNet:
    def __init__(self, cfg, hyp, ch=3):
        self.cfg, self.hyp = cfg, hyp
        gd, gw = cfg['depth_multiple'], cfg['width_multiple']
        rows = [('backbone', r) for r in cfg['backbone']] + [('neck', r) for r in cfg['fpn']] + \
               [('headers', r) for r in cfg['headers']]
        nb = len(cfg['backbone'])
        chans, self.nodes, self.shapes = [], [], {}
        for i, (part, row) in enumerate(rows):
            f, n, kind, args = row[0], row[1], row[2], list(row[3])
            tag = row[4] if len(row) > 4 else None
            args = [cfg[a] if isinstance(a, str) and a in cfg else a for a in args]
            n = max(round(n * gd), 1) if n > 1 else n
            cin = ch if i == 0 else (chans[f] if isinstance(f, int) else None)
            idx = i if part == 'backbone' else i - nb
            prefix = f'{part}.{idx}'
            node = {'i': i, 'f': f, 'kind': kind, 'prefix': prefix}
            if kind == 'Conv':
                c2 = _ceil_to(args[0] * gw, 8)
                k = args[1] if len(args) > 1 else 1
                s = args[2] if len(args) > 2 else 1
                p = args[3] if len(args) > 3 else None
                node.update(c1=cin, c2=c2, k=k, s=s, p=k // 2 if p is None else p)
                self._conv_shapes(prefix, cin, c2, k)
            elif kind == 'C3':
                c2 = _ceil_to(args[0] * gw, 8)
                shortcut = args[1] if len(args) > 1 else True
                c_ = int(c2 * 0.5)
                node.update(c1=cin, c2=c2, c_=c_, n=n, shortcut=shortcut)
                self._conv_shapes(prefix + '.cv1', cin, c_, 1)
                self._conv_shapes(prefix + '.cv2', cin, c_, 1)
                self._conv_shapes(prefix + '.cv3', 2 * c_, c2, 1)
                for j in range(n):
                    self._conv_shapes(f'{prefix}.m.{j}.cv1', c_, c_, 1)
                    self._conv_shapes(f'{prefix}.m.{j}.cv2', c_, c_, 3)
            elif kind == 'SPPF':
                c2 = _ceil_to(args[0] * gw, 8)
                c_ = cin // 2
                node.update(c1=cin, c2=c2, c_=c_, k=args[1] if len(args) > 1 else 5)
                self._conv_shapes(prefix + '.cv1', cin, c_, 1)
                self._conv_shapes(prefix + '.cv2', 4 * c_, c2, 1)
            elif kind == 'nn.Upsample':
                c2 = cin
                node.update(scale=args[1])
            elif kind == 'Concat':
                c2 = sum(chans[j] for j in f)
            elif kind == 'Detect':
                tag = tag or 'det'
                anchors, strides, nc = args[0], args[1], args[2]
                prefix = f'headers.{tag}'
                node.update(prefix=prefix, tag=tag, ch=[chans[j] for j in f], anchors=anchors,
                            strides=[float(s) for s in strides], nc=nc, no=nc + 5, na=len(anchors[0]) // 2)
                for l, c in enumerate(node['ch']):
                    self.shapes[f'{prefix}.m.{l}.weight'] = ((node['no'] * node['na'], c, 1, 1), torch.float32)
                    self.shapes[f'{prefix}.m.{l}.bias'] = ((node['no'] * node['na'],), torch.float32)
                c2 = None
            else:
                raise ValueError(f'oracle: module {kind} is outside the hot path')
            self.nodes.append(node)
            chans.append(c2)
        self.nb = nb
        self.head = [n for n in self.nodes if n['kind'] == 'Detect'][0]
        h = hyp[self.head['tag']]
        self.loss_hyp = {'box': 0.05, 'cls': 0.05, 'obj': 1.0, 'cls_pw': 1.0, 'obj_pw': 1.0, 'cls_cw': 1.0,
                         'fl_gamma': 0.0, 'iou_t': 0.20, 'anchor_t': 4.0, 'label_smoothing': 0.0}
        self.loss_hyp.update({k: h[k] for k in self.loss_hyp if k in h})
        self.nms_params = {'conf_thres': float(h.get('conf_thres', 0.15)), 'iou_thres': float(h.get('iou_thres', 0.45)),
                           'max_det': float(h.get('max_det', 300))}
        self.multi_label = bool(h['multi_label'])

    def _conv_shapes(self, p, c1, c2, k):
        self.shapes[p + '.conv.weight'] = ((c2, c1, k, k), torch.float32)
        self.shapes[p + '.bn.weight'] = ((c2,), torch.float32)
        self.shapes[p + '.bn.bias'] = ((c2,), torch.float32)
        self.shapes[p + '.bn.running_mean'] = ((c2,), torch.float32)
        self.shapes[p + '.bn.running_var'] = ((c2,), torch.float32)

    # ------------------------------------------------------------------ blocks
    def conv_bn_silu(self, sd, p, x, s, pad, training):
        # Model.freeze(layers): BatchNorm under a frozen prefix is a FrozenBatchNorm2d = the eval formula, in training too
        # (metayolo/models/utils_torch.py:180-203)
        training = training and not any(p == f or p.startswith(f + '.') for f in getattr(self, 'frozen', ()))
        y = F.conv2d(x, sd[p + '.conv.weight'], None, stride=s, padding=pad)
        y = F.batch_norm(y, sd[p + '.bn.running_mean'], sd[p + '.bn.running_var'], sd[p + '.bn.weight'],
                         sd[p + '.bn.bias'], training=training, momentum=BN_MOM, eps=BN_EPS)
        return F.silu(y)

    def _node(self, sd, nd, x, training):
        p, kind = nd['prefix'], nd['kind']
        if kind == 'Conv':
            return self.conv_bn_silu(sd, p, x, nd['s'], nd['p'], training)
        if kind == 'C3':
            a = self.conv_bn_silu(sd, p + '.cv1', x, 1, 0, training)
            for j in range(nd['n']):
                t = self.conv_bn_silu(sd, f'{p}.m.{j}.cv1', a, 1, 0, training)
                t = self.conv_bn_silu(sd, f'{p}.m.{j}.cv2', t, 1, 1, training)
                a = a + t if nd['shortcut'] else t
            b = self.conv_bn_silu(sd, p + '.cv2', x, 1, 0, training)
            return self.conv_bn_silu(sd, p + '.cv3', torch.cat((a, b), 1), 1, 0, training)
        if kind == 'SPPF':
            k = nd['k']
            a = self.conv_bn_silu(sd, p + '.cv1', x, 1, 0, training)
            p1 = F.max_pool2d(a, k, 1, k // 2)
            p2 = F.max_pool2d(p1, k, 1, k // 2)
            p3 = F.max_pool2d(p2, k, 1, k // 2)
            return self.conv_bn_silu(sd, p + '.cv2', torch.cat((a, p1, p2, p3), 1), 1, 0, training)
        if kind == 'nn.Upsample':
            return F.interpolate(x, scale_factor=float(nd['scale']), mode='nearest')
        if kind == 'Concat':
            return torch.cat(x, 1)
        raise ValueError(kind)

    def features(self, sd, x, training=False):
        """All node outputs, keyed by global node index (backbone 0.., neck nb..)."""
        outs = {}
        cur = x
        for nd in self.nodes:
            if nd['kind'] == 'Detect':
                break
            f = nd['f']
            if nd['i'] == 0:
                src = x
            elif isinstance(f, int):
                src = cur if f == -1 else outs[f]
            else:
                src = [cur if j == -1 else outs[j] for j in f]
            cur = self._node(sd, nd, src, training)
            outs[nd['i']] = cur
        return outs

    def det_logits(self, sd, feats):
        hd = self.head
        dets = []
        for l, j in enumerate(hd['f']):
            f = F.conv2d(feats[j], sd[f"{hd['prefix']}.m.{l}.weight"], sd[f"{hd['prefix']}.m.{l}.bias"])
            bs, _, ny, nx = f.shape
            dets.append(f.view(bs, hd['na'], hd['no'], ny, nx).permute(0, 1, 3, 4, 2).contiguous())
        return dets

    def decode(self, dets):
        """yolo_head.py:185-213: logits (bs,na,ny,nx,no) -> xywh px + sigmoid conf/cls."""
        hd = self.head
        preds = []
        for l, d in enumerate(dets):
            y = d.sigmoid()
            bs, na, ny, nx, no = y.shape
            stride = hd['strides'][l]
            anchor_px = torch.tensor(hd['anchors'][l], dtype=torch.float32).view(na, 2) / stride * stride
            gy, gx = torch.meshgrid(torch.arange(ny, dtype=torch.float32), torch.arange(nx, dtype=torch.float32),
                                    indexing='ij')
            grid = torch.stack((gx, gy), 2).view(1, 1, ny, nx, 2)
            xy = (y[..., 0:2] * 2. - 0.5 + grid) * stride
            wh = (y[..., 2:4] * 2.) ** 2 * anchor_px.view(1, na, 1, 1, 2)
            preds.append(torch.cat((xy, wh, y[..., 4:]), -1))
        return preds

    def outputs(self, preds):
        """yolo_head.py:301-355 (det branch) on decoded per-level preds."""
        hd = self.head
        nc, conf = hd['nc'], self.nms_params['conf_thres']
        flat = torch.cat([F.pad(p.reshape(p.shape[0], -1, hd['no']), [0, 1], value=float(l))
                          for l, p in enumerate(preds)], 1)
        res = nms_ref.nms_per_image_numpy(flat.detach().numpy(), nc, conf, self.nms_params['iou_thres'],
                                          int(self.nms_params['max_det']))
        out = []
        for r in res:
            scores = torch.from_numpy(r['scores'].copy())
            scores[:, 1:] *= scores[:, 0:1]                        # hierarchical_scores, default tree
            if self.multi_label:
                labels = scores > conf
            else:
                obj = scores[:, 0]
                if scores.shape[0]:
                    cs, cl = scores[:, 1:].max(1)
                else:
                    cs, cl = scores.new_zeros((0,)), torch.zeros((0,), dtype=torch.int64)
                labels = torch.where(cs > conf, cl + 1, torch.full_like(cl, -100))
                scores = torch.where(cs > conf, cs, obj)
            out.append({'boxes': torch.from_numpy(r['boxes'].copy()), 'scores': scores, 'labels': labels,
                        'extra': torch.from_numpy(r['extra'].copy()), 'index': torch.from_numpy(r['index'].copy())})
        return out

    # ------------------------------------------------------------------ training side
    @staticmethod
    def flatten_targets(targets, tag):
        """yolo.py:62-70 + yolo_head.py:218-223: per-image ann lists -> gts [img, cx, cy, w, h], labels."""
        anns, keep = [], []
        for i, t in enumerate(targets):
            if tag in t['anns']:
                anns.extend(t['anns'][tag])
                keep.extend([i] * len(t['anns'][tag]))
        rows, labels = [], []
        for i, a in enumerate(anns):
            b = a['boxes'].clone().clamp_(0.0, 1.0)          # xyxy2xywh(clip=True, eps=0)
            xywh = torch.stack([(b[:, 0] + b[:, 2]) / 2, (b[:, 1] + b[:, 3]) / 2, b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]], 1)
            rows.append(torch.cat([torch.full_like(xywh[:, :1], float(i)), xywh], 1))
            labels.append(a['labels'])
        return torch.cat(rows), torch.cat(labels), keep

    def match(self, shapes, gts):
        """yolo_head.py:358-417.  shapes: [(ny, nx)] per level; gts: (nt, 5) [img, cx, cy, w, h] normalised.
        Returns per level: tbox (n,4), tid (n,), (img, anchor, gj, gi), anchor wh (n,2)."""
        hd = self.head
        na, nt = hd['na'], len(gts)
        g = 0.5
        offs = torch.tensor([[0, 0], [1, 0], [0, 1], [-1, 0], [0, -1]], dtype=torch.float32) * g
        tid = torch.arange(nt, dtype=torch.float32)
        rows = torch.cat([tid[:, None], gts], 1)                               # obj, img, x, y, w, h
        ai = torch.arange(na, dtype=torch.float32).view(na, 1, 1).expand(na, nt, 1)
        rows = torch.cat([ai, rows[None].expand(na, nt, 6)], 2)                # (na, nt, 7)
        out = []
        for l, (ny, nx) in enumerate(shapes):
            anc = torch.tensor(hd['anchors'][l], dtype=torch.float32).view(na, 2) / hd['strides'][l]
            t = rows * torch.tensor([1, 1, 1, nx, ny, nx, ny], dtype=torch.float32)
            r = t[:, :, 5:7] / anc[:, None]
            ok = torch.max(r, 1. / r).max(2)[0] < self.loss_hyp['anchor_t']
            t = t[ok]
            gxy = t[:, 3:5]
            gxi = torch.tensor([nx, ny], dtype=torch.float32) - gxy
            j, k = ((gxy % 1. < g) & (gxy > 1.)).T
            lft, m = ((gxi % 1. < g) & (gxi > 1.)).T
            sel = torch.stack((torch.ones_like(j), j, k, lft, m))
            t = t.repeat((5, 1, 1))[sel]
            off = (torch.zeros_like(gxy)[None] + offs[:, None])[sel]
            gxy, gwh = t[:, 3:5], t[:, 5:7]
            gij = (gxy - off).long()
            # the reference clamps gi/gj in place through views of gij (yolo_head.py:408), so the
            # clamped cell is also the one subtracted in tbox (:409)
            gi, gj = gij[:, 0].clamp(0, nx - 1), gij[:, 1].clamp(0, ny - 1)
            gij = torch.stack((gi, gj), 1)
            a, o, b = t[:, 0].long(), t[:, 1].long(), t[:, 2].long()
            out.append((torch.cat((gxy - gij, gwh), 1), o, (b, a, gj, gi), anc[a]))
        return out

    @staticmethod
    def ciou(b1, b2, eps=1e-7):
        """utils_general.py:193-231 with xywh=True, CIoU=True on (n,4) rows."""
        x1, y1, w1, h1 = b1.chunk(4, 1)
        x2, y2, w2, h2 = b2.chunk(4, 1)
        a_x1, a_x2, a_y1, a_y2 = x1 - w1 / 2, x1 + w1 / 2, y1 - h1 / 2, y1 + h1 / 2
        b_x1, b_x2, b_y1, b_y2 = x2 - w2 / 2, x2 + w2 / 2, y2 - h2 / 2, y2 + h2 / 2
        inter = (torch.min(a_x2, b_x2) - torch.max(a_x1, b_x1)).clamp(0) * \
                (torch.min(a_y2, b_y2) - torch.max(a_y1, b_y1)).clamp(0)
        union = w1 * h1 + w2 * h2 - inter + eps
        iou = inter / union
        cw = torch.max(a_x2, b_x2) - torch.min(a_x1, b_x1)
        chh = torch.max(a_y2, b_y2) - torch.min(a_y1, b_y1)
        c2 = cw ** 2 + chh ** 2 + eps
        rho2 = ((b_x1 + b_x2 - a_x1 - a_x2) ** 2 + (b_y1 + b_y2 - a_y1 - a_y2) ** 2) / 4
        v = (4 / math.pi ** 2) * torch.pow(torch.atan(w2 / h2) - torch.atan(w1 / h1), 2)
        with torch.no_grad():
            alpha = v / (v - iou + (1 + eps))
        return iou - (rho2 / c2 + v * alpha)

    def det_loss(self, dets, targets):
        """yolo_head.py:216-229 + loss.py:190-244.  Returns (loss[1], {'box','obj','cls'})."""
        hd, hp = self.head, self.loss_hyp
        nc = hd['nc']
        gts, labels, _ = self.flatten_targets(targets, hd['tag'])
        if labels.dim() == 1:
            lab = torch.where((labels > 0) & (labels <= nc), labels, torch.zeros_like(labels))
            onehot = F.one_hot(lab, nc + 1)
        else:
            onehot = labels
        matched = self.match([d.shape[2:4] for d in dets], gts)
        balance = [4.0, 1.0, 0.4] if len(dets) == 3 else [4.0, 1.0, 0.25, 0.06, 0.02]
        lbox = torch.zeros(1)
        lobj = torch.zeros(1)
        lcls = torch.zeros(1)
        pw_c, pw_o = torch.tensor(float(hp['cls_pw'])), torch.tensor(float(hp['obj_pw']))
        cw = torch.tensor(hp['cls_cw'], dtype=torch.float32)
        for l, pi in enumerate(dets):
            tbox, tid, (b, a, gj, gi), anc = matched[l]
            tobj = torch.zeros(pi.shape[:4], dtype=pi.dtype)
            if b.shape[0]:
                ps = pi[b, a, gj, gi]
                pxy = ps[:, 0:2].sigmoid() * 2 - 0.5
                pwh = (ps[:, 2:4].sigmoid() * 2) ** 2 * anc
                iou = self.ciou(torch.cat((pxy, pwh), 1), tbox).squeeze()
                lbox = lbox + (1.0 - iou).mean()
                # loss.py:217 `tobj[b, a, gj, gi] = iou`: with several matches in one (anchor, cell) the reference keeps whichever
                # write its (parallel) index_put_ made last — run-to-run different with more than one thread.  The oracle pins the
                # one-thread behaviour: the last match in target order wins.
                lin = ((b * tobj.shape[1] + a) * tobj.shape[2] + gj) * tobj.shape[3] + gi
                order = torch.arange(lin.numel())
                last = torch.full((tobj.numel(),), -1, dtype=torch.long).scatter_reduce_(0, lin, order, reduce='amax', include_self=True)
                win = last[lin] == order
                tobj.view(-1)[lin[win]] = iou.detach().clamp(0).type(tobj.dtype)[win]
                if nc > 1:
                    tc = onehot[tid]
                    has = tc[:, 1:].sum(-1) > 0
                    if has.any():
                        tgt = tc[has][:, 1:].float()
                        tgt = tgt - (tgt - 0.5) * hp['label_smoothing']
                        bce = F.binary_cross_entropy_with_logits(ps[:, 5:][has], tgt, pos_weight=pw_c, reduction='none')
                        lcls = lcls + (bce * cw).mean()
            lobj = lobj + F.binary_cross_entropy_with_logits(pi[..., 4], tobj, pos_weight=pw_o) * balance[l]
        lbox, lobj, lcls = lbox * hp['box'], lobj * hp['obj'], lcls * hp['cls']
        bs = dets[0].shape[0]
        return (lbox + lobj + lcls) * bs, {'box': lbox.detach(), 'obj': lobj.detach(), 'cls': lcls.detach()}

    # ------------------------------------------------------------------ whole-model helpers
    def eval_forward(self, sd, x):
        feats = self.features(sd, x, training=False)
        dets = self.det_logits(sd, feats)
        preds = self.decode(dets)
        return feats, dets, preds, self.outputs(preds)

    def train_forward(self, sd, x, targets):
        """Returns loss, items, dets; running stats in `sd` are updated in place (BN momentum 0.03)."""
        feats = self.features(sd, x, training=True)
        _, _, keep = self.flatten_targets(targets, self.head['tag'])
        feats = {k: v[keep] for k, v in feats.items() if k in self.head['f']}
        dets = self.det_logits(sd, feats)
        loss, items = self.det_loss(dets, targets)
        return loss, items, dets

    def init_state(self, seed=0):
        """A full state_dict for this net from the shared synthetic-weight formula."""
        from hd_yolo_amd import synth
        return synth.synth_state_dict(self.shapes, seed=seed)


def fold_bn(sd, prefix):
    """utils_torch.py:79-99: (W', b') of conv+BN(eval) at `prefix`."""
    w = sd[prefix + '.conv.weight']
    scale = sd[prefix + '.bn.weight'] / torch.sqrt(sd[prefix + '.bn.running_var'] + BN_EPS)
    return w * scale.view(-1, 1, 1, 1), sd[prefix + '.bn.bias'] - sd[prefix + '.bn.running_mean'] * scale
