"""Detection head with the reference's constructor, attributes, buffers and state_dict keys
(reference: metayolo/models/yolo_head.py:18-23 BuffersDict, :25-112 Detect.__init__, :132-183 forward,
:185-213 compute_proposals, :216-229 compute_losses (det part), :301-355 compute_outputs (det part),
:358-417 matcher, :419-448 grids / bias init / NMS params, :473-511 hierarchical scores).

On MI355X: the per-level 1x1 convs run inside the model's HIP plan (or a head-only plan when Detect is called on
feature maps directly); decode and NMS are the `hdy_decode` / `hdy_nms_batched` kernels.  The mask branch (SURVEY.md §8 row f2;
reference :114-130, :231-275, :279-299, :320-353): the per-level seg convs are part of the model plan, roi_align and the Mask R-CNN
head run on their own kernels per call (hd_yolo_amd/maskhead.py, engine.MaskBranchFn), SegLoss is a tensor expression on the
(n, 1, 28, 28) logits.
"""
import math
from collections import OrderedDict
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from .. import LOGGER
from ... import engine as _engine
from ... import ops as _ops
from .layers import Conv
from .loss import DetLoss, SegLoss
from .mask_rcnn import MaskRCNNHeads, MaskRCNNPredictor
from .utils_general import nms_per_image, paired_box_iou, xywh2xyxy, xyxy2xywh
from .utils_torch import one_hot_labels


class BuffersDict(nn.Module):
    def __init__(self, x: Dict[str, Optional[torch.Tensor]] = {}):
        super().__init__()
        for k, v in x.items():
            self.register_buffer(k, v)


class _FeatureList(list):
    """pseudo-backbone for a head-only plan (Detect called on given feature maps)"""
    save = []


class Detect(nn.Module):
    def __init__(self, ch: List[int], anchors: List[List[int]], strides: List[int], nc: int, masks: Dict[int, int] = {},
                 dim_reduced: int = 256, mask_output_size: int = 28, multi_label: bool = False,
                 nms_params: Dict[str, float] = {}, loss_hyp: Dict[str, float] = {}, default_input_size: Optional[int] = 640,
                 is_scripting: bool = False):
        super().__init__()
        assert len(ch) == len(anchors) == len(strides), 'ch, anchors, strides should have same length.'
        self.ch, self.nl, self.nc, self.no = ch, len(ch), nc, nc + 5
        self.na = len(anchors[0]) // 2
        self.default_input_size = default_input_size
        self.descendants: Dict[int, List[int]] = {}
        self.get_descendants(self.build_hierarchical_tree())

        strides_t = torch.tensor(strides).float()
        anchors_t = torch.tensor(anchors).float().view(len(anchors), -1, 2) / strides_t.view(-1, 1, 1)   # in grid units
        self.anchors = nn.ModuleList([])
        for s, a in zip(strides_t, anchors_t):
            if default_input_size is not None:
                n = int(default_input_size / s)
                grid, anchor_grid = self._make_grid(a, s, n, n)
            else:
                grid, anchor_grid = None, None
            self.anchors.append(BuffersDict({'stride': s, 'anchor': a, 'grid': grid, 'anchor_grid': anchor_grid}))

        self.m = self.build_det_layers()
        self.initialize_biases()
        if not is_scripting:
            self.det_loss = DetLoss(self.nc, self.nl, loss_hyp, ssi=0)
        self.nms_params: Dict[str, float] = self.get_nms_params(nms_params)
        self.multi_label: bool = multi_label

        mask_indices = torch.tensor([masks.get(i, 0) for i in range(self.nc + 1)])
        self.nc_masks = mask_indices.max().item() + 1
        self.register_buffer('mask_indices', mask_indices)
        self.dim_reduced = dim_reduced
        if self.nc_masks > 0:
            self.mask_output_size = mask_output_size
            self.seg, self.seg_h = self.build_seg_layers()
            self.aligned = False                      # the reference's ROI_ALIGN global (yolo_head.py:15)
            self.seg_loss = None if is_scripting else SegLoss(loss_hyp)
        else:
            self.mask_output_size = None
            self.seg, self.seg_h, self.seg_loss = None, None, None

    # ------------------------------------------------------------------ construction helpers
    def build_det_layers(self):
        return nn.ModuleList(nn.Conv2d(c, self.no * self.na, 1) for c in self.ch)

    def build_seg_layers(self):
        """One 3x3 Conv (-> dim_reduced) per level, top-down like the reference (yolo_head.py:123-128), and the Mask R-CNN head."""
        seg = nn.ModuleList(Conv(self.ch[i], self.dim_reduced, kernel_size=3, act=True) for i in range(self.nl - 1, -1, -1))
        seg_h = nn.Sequential(OrderedDict([
            ('maskrcnn_heads', MaskRCNNHeads(self.dim_reduced, (256, 256, 256, 256), 1)),
            ('maskrcnn_preds', MaskRCNNPredictor(256, 256, self.nc_masks)),
        ]))
        return seg, seg_h

    def _make_grid(self, anchor, stride, nx: int = 20, ny: int = 20):
        d, t = anchor.device, anchor.dtype
        yv, xv = torch.meshgrid(torch.arange(ny, device=d, dtype=t), torch.arange(nx, device=d, dtype=t), indexing='ij')
        return torch.stack((xv, yv), 2), anchor * stride

    def initialize_biases(self, cf=None):
        """obj prior of 8 objects per 640 px image, cls prior 0.6 / nc (focal-loss paper, section 3.3)."""
        for mi, buf in zip(self.m, self.anchors):
            b = mi.bias.view(self.na, -1)
            b.data[:, 4] += math.log(8 / (640 / buf.stride) ** 2)
            b.data[:, 5:] += math.log(0.6 / (self.nc - 0.999999)) if cf is None else torch.log(cf / cf.sum())
            mi.bias = torch.nn.Parameter(b.view(-1), requires_grad=True)

    def _print_biases(self):
        for mi in self.m:
            b = mi.bias.detach().view(self.na, -1).T
            LOGGER.info(('%6g Conv2d.bias:' + '%10.3g' * 6) % (mi.weight.shape[1], *b[:5].mean(1).tolist(), b[5:].mean()))

    def get_nms_params(self, args={}):
        defaults = {'conf_thres': 0.15, 'iou_thres': 0.45, 'max_det': 300}
        return {k: float(args.get(k, v)) for k, v in defaults.items()}

    def build_hierarchical_tree(self):
        return {0: {c: None for c in range(1, self.nc + 1)}}

    def get_descendants(self, node: Optional[Dict] = None):
        res: List[int] = []
        if node is not None:
            for k, v in node.items():
                res.append(k)
                below = self.get_descendants(v)
                if below:
                    self.descendants[k] = below
                    res += below
        return res

    def _score_pairs(self, dev):
        """(child, parent) score columns in the order hierarchical_scores applies them, as an int32 device tensor (cached per device)"""
        key = (str(dev), tuple((k, tuple(v)) for k, v in self.descendants.items()))
        cache = self.__dict__.setdefault('_pairs', {})
        if cache.get('key') != key:
            pairs = [(c, k) for k, v in self.descendants.items() for c in v]
            cache['key'], cache['t'] = key, torch.tensor(pairs, dtype=torch.int32, device=dev).reshape(-1, 2)
        return cache['t']

    def hierarchical_scores(self, x: torch.Tensor) -> torch.Tensor:
        """In place: every node's score is multiplied by its ancestors' (default tree: cls *= obj)."""
        for k, v in self.descendants.items():
            x[:, v] *= x[:, k:k + 1]
        return x

    # ------------------------------------------------------------------ forward
    def forward(self, x: Dict[int, torch.Tensor], targets=None, compute_masks: bool = True):
        """x: {layer index: NCHW feature map}.  Runs the level convs on a head-only HIP plan, then the shared tail."""
        if self.training and torch.is_grad_enabled():
            raise RuntimeError('Detect.forward on bare features is forward-only: train through Model (the det convs are part of '
                               'the model plan and its backward)')
        f = self.f if isinstance(self.f, (list, tuple)) else [self.f]
        return self.forward_dets(self._det_convs({j: x[j] for j in f}), targets, compute_masks=compute_masks)

    def _det_convs(self, feats):
        eng = self.__dict__.get('_hdy_engine')
        if eng is None:
            eng = _engine.Engine(None, None, self)
            object.__setattr__(self, '_hdy_engine', eng)
        plan = eng.plan_for_features(feats, _engine.compute_dtype(self, next(iter(feats.values()))))
        return list(plan.run_forward_features(feats))

    def forward_dets(self, dets: List[torch.Tensor], targets=None, compute_masks: bool = True, mask_ctx=None):
        """Tail of Detect.forward given the per-level logits (bs, na, ny, nx, no) fp32.  mask_ctx = (engine, plan, dtype) of the
        launch plan whose mask feature maps the mask branch reads (None: no mask branch in this call)."""
        if self.training:
            assert targets is not None
            want_loss, want_out = True, False
        else:
            want_loss, want_out = targets is not None, True
        compute_masks = bool(compute_masks) and self.nc_masks > 0 and mask_ctx is not None
        if want_out and not want_loss:
            # plain inference: all levels are decoded straight into the (bs, N, no + 1) tensor the NMS kernel reads
            return {}, self.compute_outputs(self.decode_all(dets), mask_ctx, compute_masks=compute_masks)
        preds = self.compute_proposals(dets) if (want_out or (want_loss and compute_masks)) else []
        losses = self.compute_losses(dets, preds, mask_ctx, targets, compute_masks=compute_masks) if want_loss else {}
        outputs = self.compute_outputs(preds, mask_ctx, compute_masks=compute_masks) if want_out else []
        return losses, outputs

    def anchor_px(self, i):
        buf = self.anchors[i]
        return (buf.anchor * buf.stride).flatten().tolist()

    def compute_proposals(self, dets: List[torch.Tensor]) -> List[torch.Tensor]:
        """Per level (bs, na, ny, nx, no): xy = (sigmoid*2 - 0.5 + grid)*stride, wh = (sigmoid*2)^2 * anchor px, rest sigmoid."""
        preds = []
        for i, d in enumerate(dets):
            bs, na, ny, nx, no = d.shape
            out = torch.empty((bs, na * ny * nx, no + 1), dtype=torch.float32, device=d.device)
            _ops.decode_level(d, self._anchor_px_cached(i), self._stride_cached(i), out, 0, i)
            preds.append(out[..., :no].view(bs, na, ny, nx, no))
        return preds

    def _stride_cached(self, i):
        """Stride of level i as a python float (the buffer lives on the GPU: float(buffer) is a device-to-host sync per call).
        Cached against the buffer's version counter, which load_state_dict bumps."""
        buf = self.anchors[i].stride
        cache = self.__dict__.setdefault('_strides', {})
        if cache.get(i, (None, None))[0] != (id(buf), buf._version):
            cache[i] = ((id(buf), buf._version), float(buf))
        return cache[i][1]

    def _anchor_px_cached(self, i):
        bufs = self.anchors[i]
        key = (id(bufs.anchor), bufs.anchor._version, id(bufs.stride), bufs.stride._version)
        cache = self.__dict__.setdefault('_apx', {})
        if cache.get(i, (None, None))[0] != key:
            cache[i] = (key, self.anchor_px(i))
        return cache[i][1]

    def decode_all(self, dets: List[torch.Tensor]) -> torch.Tensor:
        """All levels decoded straight into one (bs, sum na*ny*nx, no+1) tensor with the level id in the last column
        (what compute_outputs feeds to NMS; replaces the reference's per-level pad + cat)."""
        bs = dets[0].shape[0]
        rows = [d.shape[1] * d.shape[2] * d.shape[3] for d in dets]
        out = torch.empty((bs, sum(rows), self.no + 1), dtype=torch.float32, device=dets[0].device)
        off = 0
        for i, d in enumerate(dets):
            _ops.decode_level(d, self._anchor_px_cached(i), self._stride_cached(i), out, off, i)
            off += rows[i]
        return out

    def compute_outputs(self, preds, features=(), compute_masks: bool = False) -> List[Dict[str, torch.Tensor]]:
        """preds: list of decoded levels (bs, na, ny, nx, no) or the already concatenated (bs, N, no+1) tensor."""
        if isinstance(preds, (list, tuple)):
            flat = torch.cat([torch.nn.functional.pad(p.reshape(p.shape[0], -1, self.no), [0, 1], value=float(i))
                              for i, p in enumerate(preds)], 1)
        else:
            flat = preds
        conf = self.nms_params['conf_thres']
        max_det = int(self.nms_params['max_det'])
        # one NMS launch for the batch, then the score / label logic ONCE on the padded (bs * max_det, 1 + nc) rows (rows past an
        # image's count are scratch and sliced away): the reference's per-image loop (yolo_head.py:313-353) costs ~10 tiny launches
        # per tile, 17 ms of a 96 ms batch of 128 1024x1024 tiles
        bs = flat.shape[0]
        if bs == 0:
            return []
        res = _ops.nms_batched(flat.float().contiguous(), self.nc, conf, self.nms_params['iou_thres'], max_det, min_wh=2.0, class_aware=False)
        # hierarchical scores, best class / objectness fallback and labels for every kept box, compacted over the batch by ONE launch
        # (hdy_det_outputs) issued before the sync; the host then splits three tensors instead of slicing 3 x bs padded ones
        boxes_c, scores_c, labels_c = _ops.det_outputs(res, self.nc, conf, self._score_pairs(flat.device), self.multi_label)
        n_keep = res['n_keep'].tolist()                    # the one D2H sync of the batch
        total = sum(n_keep)
        results = [{'boxes': b, 'scores': s, 'labels': l} for b, s, l in
                   zip(boxes_c[:total].split(n_keep), scores_c[:total].split(n_keep), labels_c[:total].split(n_keep))]
        if compute_masks and sum(n_keep) > 0 and not self.multi_label:
            self.attach_masks(results, res, n_keep, features)
        return results

    def attach_masks(self, results, res, n_keep, mask_ctx):
        """multiscale_roi_align over the detections (each from the level that produced it) -> Mask R-CNN head -> sigmoid -> the
        channel of the detection's mask label; 28 x 28 masks in box coordinates (reference: yolo_head.py:320-353, :279-299)."""
        from ...maskhead import MaskHeadRun
        engine, plan, dtype = mask_ctx
        dev = res['boxes'].device
        img = torch.cat([torch.full((n,), float(b), device=dev) for b, n in enumerate(n_keep)])
        boxes = torch.cat([res['boxes'][b, :n] for b, n in enumerate(n_keep)])
        levels = torch.cat([res['extra'][b, :n, 0] for b, n in enumerate(n_keep)]).long()
        rois = torch.cat([img[:, None], boxes], 1)
        P = self.mask_output_size // 2
        feats = plan.mask_features()
        parts, pos = [], []
        for l in range(self.nl):
            sel = (levels == l).nonzero().flatten()
            pos.append(sel)
            parts.append(_ops.roi_align(feats[l], rois[sel], 1.0 / self._stride_cached(l), P, 2, self.aligned))
        order = torch.empty(len(rois), dtype=torch.long, device=dev)
        order[torch.cat(pos)] = torch.arange(len(rois), device=dev)
        logits = MaskHeadRun(self.seg_h, dtype).forward(torch.cat(parts)[order].contiguous())        # (R, 28, 28, nc_masks) fp32
        probs = logits.sigmoid().permute(0, 3, 1, 2).split(n_keep, dim=0)
        for r, m in zip(results, probs):
            if len(m):
                mask_labels = self.mask_indices[r['labels'].clamp(min=0)]
                r['masks'] = m[torch.arange(len(m), device=dev), mask_labels][:, None]
                r['masks'][mask_labels < 0] = 0

    # ------------------------------------------------------------------ training side
    def fused_loss_ok(self):
        """The single-launch-sequence loss (csrc/loss.hip) covers the default DetLoss: BCE (no focal), fixed balance."""
        import os
        dl = self.det_loss
        return (os.environ.get('HDY_FUSED_LOSS', '1') != '0' and not dl.autobalance and dl.hyp['fl_gamma'] == 0 and dl.gr == 1.0
                and not dl.sort_obj_iou and self.nc <= 128)

    def flatten_targets(self, targets, dev, fused=False):
        """Per-image ann dicts -> gts (nt,5) [img, cx, cy, w, h] and one-hot labels (nt, nc+1), built once per batch.
        fused=True (device path, plain integer labels): one kernel, and the second result is tcls (nt, nc) float instead — what
        hdy_det_loss takes; the (nt, nc+1) table is only built by callers that need it."""
        boxes = [t['boxes'] for t in targets]
        if boxes:
            torch._foreach_clamp_min_(boxes, 0.0)      # the reference's xyxy2xywh(clip=True) clamps the caller's boxes too
            torch._foreach_clamp_max_(boxes, 1.0)
        counts = [int(b.shape[0]) for b in boxes]

        def up(t):
            # host tensors go up through pinned memory, non-blocking: a pageable .to(dev) makes the host wait for everything queued so
            # far (the previous step's backward), after which the GPU idles until the forward launches arrive: 16.5 -> 15.9 ms per bench step
            if dev.type == 'cuda' and t.device.type == 'cpu':
                return t.pin_memory().to(dev, non_blocking=True)
            return t.to(dev)

        allb = up(torch.cat(boxes)) if boxes else torch.zeros((0, 4), device=dev)
        img = up(torch.repeat_interleave(torch.arange(len(boxes), dtype=allb.dtype), torch.tensor(counts)))       # one small upload
        labs = [t['labels'] for t in targets]
        plain = all(l.dim() == 1 for l in labs)
        if fused and plain and dev.type == 'cuda' and allb.dtype == torch.float32 and len(labs) > 0:
            from hd_yolo_amd import ops
            return ops.det_targets(allb.contiguous(), img, up(torch.cat(labs)).long(), self.nc)
        gts = torch.stack([img, (allb[:, 0] + allb[:, 2]) / 2, (allb[:, 1] + allb[:, 3]) / 2, allb[:, 2] - allb[:, 0],
                           allb[:, 3] - allb[:, 1]], 1)
        if plain:
            gt_labels = one_hot_labels(up(torch.cat(labs)), self.nc)
        else:
            gt_labels = up(torch.cat([one_hot_labels(l, self.nc) if l.dim() == 1 else l for l in labs]))
        if fused:
            return gts.contiguous(), gt_labels[:, 1:].float().contiguous()
        return gts, gt_labels

    def fused_losses(self, engine, x, dtype, targets, compute_masks=False):
        want_masks = compute_masks and self.nc_masks > 0
        if want_masks:
            gts, gt_labels = self.flatten_targets(targets, x.device)
            gts, tcls = gts.contiguous(), gt_labels[:, 1:].float().contiguous()
        else:
            gts, tcls = self.flatten_targets(targets, x.device, fused=True)
        plan, loss, items = engine.forward_fused_loss(x, dtype, self, gts, tcls)
        mask_loss = None
        if want_masks:
            mask_loss = self.mask_losses_device((engine, plan, dtype), targets, gts, gt_labels)
        if mask_loss is None:
            mask_loss = torch.zeros_like(loss)
        return plan, {'det_loss': loss, 'mask_loss': mask_loss,
                      'loss_items': {'box': items[0:1], 'obj': items[1:2], 'cls': items[2:3], 'mask': mask_loss.detach()}}

    def compute_losses(self, dets, preds, features, targets, compute_masks=False):
        """Target rows [img, cx, cy, w, h] + one-hot labels -> matcher -> DetLoss.  The reference builds the rows image by
        image (yolo_head.py:218-223); here the per-image box tensors are clamped in place with one multi-tensor call (the
        reference's xyxy2xywh(clip=True) also clamps the caller's boxes) and everything else runs once on the concatenation."""
        gts, gt_labels = self.flatten_targets(targets, dets[0].device)
        tbox, tids, indices, anchors = self.matcher(dets, gts)
        tcls = [gt_labels[i] for i in tids]
        det_loss, items = self.det_loss(dets, tcls, tbox, indices, anchors)
        mask_loss = self.mask_losses(preds, features, targets, gts, tids, indices, tcls) if compute_masks else None
        if mask_loss is None:
            mask_loss = torch.zeros_like(det_loss)
        return {'det_loss': det_loss, 'mask_loss': mask_loss, 'loss_items': {**items, 'mask': mask_loss.detach()}}

    def mask_losses(self, preds, mask_ctx, targets, gts, tids, indices, tcls):
        """Mask loss of the matched cells (reference: yolo_head.py:231-275): per level, roi_align of the mask feature map over the
        GROUND-TRUTH boxes of the matched targets; of all cells matched to one target only the one whose predicted box has the
        best IoU with the truth, and only if that IoU >= 0.8, goes through the Mask R-CNN head; SegLoss against the 28 x 28 target."""
        engine, plan, dtype = mask_ctx
        dev = gts.device
        props, gt_props, obj_ids, rois = [], [], [], []
        for i, buf in enumerate(self.anchors):
            v = plan.mask_vals[i]
            b, a, gj, gi = indices[i]
            boxes = xywh2xyxy(preds[i][b, a, gj, gi, :4].detach())
            gt_boxes = xywh2xyxy(gts[tids[i]][:, 1:] * gts.new([v.w, v.h, v.w, v.h]) * buf.stride)
            props.append(boxes)
            gt_props.append(gt_boxes)
            obj_ids.append(tids[i])
            rois.append(torch.cat([b[:, None].to(gt_boxes.dtype), gt_boxes], -1))
        sizes = [len(o) for o in obj_ids]
        props, gt_props, obj_ids = torch.cat(props), torch.cat(gt_props), torch.cat(obj_ids)
        if not len(obj_ids):
            return None
        # torch_scatter.scatter_max(box_ious, obj_ids): per target its best IoU and the FIRST row attaining it
        ious = paired_box_iou(props, gt_props)
        nobj = int(gts.shape[0])
        best = torch.zeros(nobj, dtype=ious.dtype, device=dev).scatter_reduce_(0, obj_ids, ious, 'amax', include_self=False)
        rows = torch.arange(len(ious), device=dev)
        hit = ious == best[obj_ids]
        arg = torch.full((nobj,), len(ious), dtype=torch.long, device=dev).scatter_reduce_(0, obj_ids[hit], rows[hit], 'amin')
        present = torch.zeros(nobj, dtype=torch.bool, device=dev).index_fill_(0, obj_ids, True)
        keep = arg[(best >= 0.8) & present]
        if not len(keep):
            return None
        # roi_align only what is kept; rows are concatenated level by level, `order` puts them in `keep` order
        level = torch.repeat_interleave(torch.arange(self.nl, device=dev), torch.tensor(sizes, device=dev))
        all_rois = torch.cat(rois)
        klev = level[keep]
        by_level = [all_rois[keep[klev == l]] for l in range(self.nl)]
        pos = torch.cat([(klev == l).nonzero().flatten() for l in range(self.nl)])       # position in `keep` of each concatenated row
        order = torch.empty_like(pos)
        order[pos] = torch.arange(len(pos), device=dev)
        mask_logits = _engine.MaskBranchFn.apply(engine.mask_token, engine, plan, self, by_level, order, dtype)
        mask_targets = torch.cat([t['masks'] for t in targets]).to(dev)[obj_ids[keep], None] * 1.0
        gt_labels = torch.cat(tcls)[keep]
        hier = (gt_labels * torch.arange(self.nc + 1, device=dev)).max(-1)[1]
        return self.seg_loss(mask_logits, mask_targets, self.mask_indices[hier])

    def mask_losses_device(self, mask_ctx, targets, gts, gt_labels):
        """mask_losses with the selection on the device (hdy_mask_select on the plan's logits: the same candidates as the loss kernel's matcher,
        the decode kernel's boxes, the IoU of paired_box_iou) and ONE device-to-host sync for the row counts; the tensor-expression version
        (matcher + decode of every level + scatter_reduce + per-level boolean indexing) made ~20."""
        engine, plan, dtype = mask_ctx
        dev = gts.device
        for i, v in enumerate(plan.mask_vals):
            d = plan.det_units[i].x
            if (v.w, v.h) != (d.w, d.h):
                raise RuntimeError('mask feature maps and detection levels differ in size')
        masks = torch.cat([t['masks'] for t in targets])
        if masks.device.type == 'cpu' and dev.type == 'cuda':          # pinned, non-blocking: a pageable upload waits for everything queued so far
            masks = masks.pin_memory().to(dev, non_blocking=True)
        else:
            masks = masks.to(dev)
        apx = [v for i in range(self.nl) for v in self._anchor_px_cached(i)]
        counts, keep_t, rois, order = plan.fused_loss(self).mask_select(gts, apx, [self._stride_cached(i) for i in range(self.nl)])
        counts = counts.tolist()                                        # the one sync
        nk = counts[0]
        if nk == 0:
            return None
        by_level = [rois[l, :counts[1 + l]] for l in range(self.nl)]
        kt = keep_t[:nk]
        mask_logits = _engine.MaskBranchFn.apply(engine.mask_token, engine, plan, self, by_level, order[:nk], dtype)
        mask_targets = masks[kt, None] * 1.0
        hier = (gt_labels[kt] * self._const(('classes',), dev, lambda: torch.arange(self.nc + 1))).max(-1)[1]
        return self.seg_loss(mask_logits, mask_targets, self.mask_indices[hier])

    def _const(self, key, dev, make):
        """small constant tensors are uploaded once per device, not once per step (each upload is a host sync)"""
        cache = self.__dict__.setdefault('_consts', {})
        k = (key, str(dev))
        if k not in cache:
            cache[k] = make().to(dev)
        return cache[k]

    def matcher(self, p, gts):
        """Assign each ground-truth box (img, cx, cy, w, h; normalised) to anchors whose w/h ratio is within anchor_t and to
        the cell containing its centre plus the up-to-two nearest neighbour cells.  Returns per level
        (tbox [dx, dy, w, h in grid units], object ids, (img, anchor, gj, gi), anchor wh)."""
        dev = gts.device
        na, nt = self.na, len(gts)
        rows = torch.cat([torch.arange(nt, device=dev, dtype=gts.dtype)[:, None], gts[:, :6]], 1)        # obj, img, x, y, w, h
        ai = torch.arange(na, device=dev, dtype=gts.dtype).view(na, 1, 1).expand(na, nt, 1)
        rows = torch.cat([ai, rows[None].expand(na, nt, rows.shape[1])], 2)                               # (na, nt, 7)
        g = 0.5
        shifts = self._const(('shifts',), dev, lambda: torch.tensor([[0, 0], [1, 0], [0, 1], [-1, 0], [0, -1]], dtype=gts.dtype) * g)
        tbox, tids, indices, anch = [], [], [], []
        for i, buf in enumerate(self.anchors):
            anc = buf.anchor.to(dev)
            ny, nx = p[i].shape[2:4]
            t = rows * self._const(('gain', nx, ny), dev, lambda: torch.tensor([1, 1, 1, nx, ny, nx, ny], dtype=gts.dtype))
            if nt:
                ratio = t[:, :, 5:7] / anc[:, None]
                t = t[torch.max(ratio, 1. / ratio).max(2)[0] < self.det_loss.hyp['anchor_t']]
                gxy = t[:, 3:5]
                inv = self._const(('wh', nx, ny), dev, lambda: torch.tensor([nx, ny], dtype=gts.dtype)) - gxy
                j, k = ((gxy % 1. < g) & (gxy > 1.)).T
                l, m = ((inv % 1. < g) & (inv > 1.)).T
                sel = torch.stack((torch.ones_like(j), j, k, l, m))
                t = t.repeat((5, 1, 1))[sel]
                off = (torch.zeros_like(gxy)[None] + shifts[:, None])[sel]
            else:
                t, off = rows[0], 0
            gxy, gwh = t[:, 3:5], t[:, 5:7]
            cell = (gxy - off).long()
            gi, gj = cell[:, 0].clamp(0, nx - 1), cell[:, 1].clamp(0, ny - 1)
            cell = torch.stack((gi, gj), 1)          # the reference clamps through views, so tbox uses the clamped cell
            a, o, b = t[:, 0].long(), t[:, 1].long(), t[:, 2].long()
            indices.append((b, a, gj, gi))
            tbox.append(torch.cat((gxy - cell, gwh), 1))
            anch.append(anc[a])
            tids.append(o)
        return tbox, tids, indices, anch

    # ------------------------------------------------------------------ whole-slide helpers (yolo_head.py:450-471)
    def merge_outputs(self, r):
        boxes = torch.cat([x['boxes'] + x['boxes'].new([x['roi'][0], x['roi'][1], x['roi'][0], x['roi'][1]]) for x in r])
        res = {'boxes': boxes, 'labels': torch.cat([x['labels'] for x in r]), 'scores': torch.cat([x['scores'] for x in r])}
        if 'masks' in r[0]:
            res['masks'] = torch.cat([x['masks'] for x in r])
        return res

    def rescale_outputs(self, r, scale=1.0):
        if scale != 1.0:
            r['boxes'] *= scale
        return r
