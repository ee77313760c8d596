"""PanopticSeg: semantic-segmentation header over a feature pyramid (reference: hnet/segmentation/panoptic_seg.py:3-43).

config keys kept from the reference: in_channels, num_classes, feature_maps (ordered {name: name}, finest level first), scale_factor,
resize_mode ('bilinear'), class_weight.  forward(features, image_size, roi_size, targets=None) -> (probabilities split per image, losses):
    features   {name: (N, C, h, w)} on the GPU
    targets    per image a list of annotations {'roi': (4,) xyxy in image pixels, 'masks': (num_classes, H, W) float}; this build takes
               whole-tile rois only — for those the reference's roi_align(aligned=True, output = map size) samples every bin once, at
               the pixel centre, i.e. it is the identity (hnet/utils.py:143-154)
    losses     {'soft_iou_loss': 1 + criterion(probabilities, masks)} (reference :40) with criterion = -weighted soft dice
`SoftDiceLoss` is referenced but not defined anywhere in the reference repository; it is restated here from the repository's own dice
(metayolo/models/utils_general.py:268-280, mask_iou(factor=0)): dice[n][c] = 2*sum(t*p) / sum(t + p) over the pixels,
criterion = - sum_c w_c * mean_n dice[n][c] / sum_c w_c."""
import torch

from ... import _lib, ops
from ...segrun import PackCache, PanopticRun
from .utils_seg import PanopticFeatureConnector

__all__ = ['PanopticSeg', 'SoftDiceLoss']


class SoftDiceLoss(torch.nn.Module):
    def __init__(self, class_weight=None):
        super().__init__()
        self.class_weight = None if class_weight is None else [float(w) for w in class_weight]

    def weights(self, nc, device):
        if self.class_weight is None:
            return None
        assert len(self.class_weight) == nc
        return torch.tensor(self.class_weight, dtype=torch.float32, device=device)

    def forward(self, probs, masks):
        """probabilities (N, nc, H, W), masks (N, nc, H, W) -> -weighted mean dice (tensor expressions; the fused kernel is hdy_softdice)"""
        prod, plus = (masks * probs).sum((2, 3)), (masks + probs).sum((2, 3))
        dice = (2 * prod / plus).mean(0)
        w = torch.ones_like(dice) if self.class_weight is None else self.weights(dice.numel(), dice.device)
        return -(dice * w).sum() / w.sum()


class _Runs(dict):
    """executors (device buffers, recorded launch lists) of the module OBJECT they were made for: a copied or unpickled module starts without"""

    def __deepcopy__(self, memo):
        return _Runs()

    def __reduce__(self):
        return (_Runs, ())


class _SegFn(torch.autograd.Function):
    """connector + class head + soft dice as ONE autograd node over the pyramid features: forward -> loss (1,); backward -> feature
    gradients (NCHW-shaped, channels-last) and the header's parameter gradients written through `grad_of`."""

    @staticmethod
    def forward(ctx, seg, run, grad_of, masks, *feats):
        nhwc = [f.permute(0, 2, 3, 1) for f in feats]
        nhwc = [f if f.dtype == run.dtype else f.to(run.dtype) for f in nhwc]
        logits = run.forward([f if ops_is_nhwc(f) else f.contiguous() for f in nhwc], out_size=masks.shape[-2:], train=True)
        cw = seg.criterion.weights(seg.config['num_classes'], logits.device)
        low_w = run.low_shape[2]
        ctx.w_reduced = tuple(run.out_size) != tuple(run.low_shape[1:3]) and ops.softdice_wgrad_ok(logits, masks.shape[1], low_w)
        if ctx.w_reduced:          # loss + gradient already reduced along W: the full-resolution gradient tensor is never written
            loss, dl = ops.softdice_wgrad(logits, masks, cw, low_w, bufs=run.__dict__.setdefault('_loss_bufs', {}))
        else:
            loss, dl = ops.softdice(logits, masks, cw, want_grad=True)
        ctx.run, ctx.dl, ctx.grad_of, ctx.dtypes, ctx.gen = run, dl, grad_of, [f.dtype for f in feats], run.gen
        # `loss` (and `dl`) live in the run's persistent buffers, the same storage every step: the caller gets its own element (a history of
        # `loss.detach()` kept across steps must not all read the latest value)
        return loss.clone()

    @staticmethod
    def backward(ctx, g):
        ctx.run.check_generation(ctx.gen)
        dfeats = ctx.run.backward(ctx.dl, ctx.grad_of, scale=g, w_reduced=ctx.w_reduced)          # the upstream factor goes in at the connector's resolution
        return (None, None, None, None) + tuple(d.permute(0, 3, 1, 2).to(t) for d, t in zip(dfeats, ctx.dtypes))


def ops_is_nhwc(t):
    try:
        ops.nhwc(t)
        return True
    except AssertionError:
        return False


def _stack_masks(ms):
    """torch.stack(ms).float().contiguous() (reference: panoptic_seg.py:36) — without the copy when the loader already holds the per-image
    masks as consecutive fp32 slices of one batch tensor (a 79 MB device copy per step at 16 x 3 x 1280 x 1280)."""
    m0 = ms[0]
    if (m0.dtype == torch.float32 and m0.is_contiguous() and
            all(m.dtype == m0.dtype and m.shape == m0.shape and m.is_contiguous() and m.device == m0.device and
                m.untyped_storage().data_ptr() == m0.untyped_storage().data_ptr() and
                m.storage_offset() == m0.storage_offset() + i * m0.numel() for i, m in enumerate(ms))):
        return m0.as_strided((len(ms),) + tuple(m0.shape), (m0.numel(),) + tuple(m0.stride()), m0.storage_offset())
    return torch.stack(ms).float().contiguous()


class PanopticSeg(torch.nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.config['featmap_names'] = list(self.config['feature_maps'].keys())
        self.connector = PanopticFeatureConnector(config['in_channels'], config['in_channels'], config['feature_maps'], mode='bilinear')
        layers = [torch.nn.Conv2d(config['in_channels'], config['num_classes'], kernel_size=1), torch.nn.Softmax2d()]
        scale_factor = config.get('scale_factor')
        if scale_factor is not None and scale_factor != 1:
            if config.get('resize_mode', 'bilinear') != 'bilinear':
                raise _lib.HdyError('PanopticSeg: only bilinear resizing has a HIP kernel')
            layers = [torch.nn.Upsample(scale_factor=scale_factor, mode='bilinear', align_corners=True)] + layers
        self.layers = torch.nn.Sequential(*layers)
        self.criterion = SoftDiceLoss(config.get('class_weight'))

    def class_conv(self):
        return [m for m in self.layers if isinstance(m, torch.nn.Conv2d)][0]

    def _run(self, dtype):
        """one executor per arithmetic type, kept across calls: it owns the tapes of the training step (hd_yolo_amd/segrun.py)"""
        cache = self.__dict__.setdefault('_hdy_pack', PackCache())
        runs = self.__dict__.setdefault('_hdy_runs', _Runs())
        run = runs.get(dtype)
        if run is None or run.class_conv is not self.class_conv():
            run = runs[dtype] = PanopticRun(self.connector, self.class_conv(), dtype, cache)
        return run

    def _param_grad(self, p):
        """default gradient sink: the parameter's own .grad (an engine passes its flat store's views instead)"""
        if p.grad is None:
            p.grad = torch.zeros_like(p, dtype=torch.float32)
        return p.grad

    def forward(self, features, image_size=None, roi_size=None, targets=None, grad_of=None, dtype=None):
        if self.training and targets is None:
            raise ValueError('In training mode, targets should be passed')
        feats = [features[k] for k in self.connector.layers.keys()]
        ops.require_gpu(feats[0])
        N, _, h, w = feats[0].shape
        if dtype is None:
            dtype = torch.bfloat16 if (feats[0].dtype in (torch.bfloat16, torch.float16) or torch.is_autocast_enabled()) else torch.float32
        s = self.config.get('scale_factor') or 1
        out_size = (int(h * s), int(w * s))
        losses, counts = {}, [1] * N
        run = self._run(dtype)
        if targets is not None:
            counts = [len(t) for t in targets]
            anns = [a for t in targets for a in t]
            if counts != [1] * N:
                raise _lib.HdyError('PanopticSeg on this path takes one whole-tile roi per image')
            if image_size is not None:
                H, W = (image_size, image_size) if isinstance(image_size, int) else tuple(image_size)
                for a in anns:
                    if 'roi' in a and [float(v) for v in a['roi']] != [0.0, 0.0, float(W), float(H)]:
                        raise _lib.HdyError('PanopticSeg on this path takes whole-tile rois only (roi == [0, 0, W, H])')
            masks = _stack_masks([a['masks'] for a in anns])
            if tuple(masks.shape[-2:]) != out_size:
                # reference (panoptic_seg.py:13-19, :37-39): Upsample(scale_factor) -> conv -> Softmax2d, THEN the probabilities are
                # interpolated to the mask size.  This path resizes the logits once and applies the softmax last, which is the same
                # function only when feature size x scale_factor == mask size (softmax and interpolation do not commute).
                raise _lib.HdyError(f'PanopticSeg: masks of {tuple(masks.shape[-2:])} do not match feature size x scale_factor = {out_size}; '
                                    'the HIP path covers the case where they agree (set scale_factor accordingly)')
            if torch.is_grad_enabled() and self.training:
                loss = _SegFn.apply(self, run, grad_of or self._param_grad, masks, *feats)
            else:
                logits = self._logits(run, feats, masks.shape[-2:], dtype)
                loss, _ = ops.softdice(logits, masks, self.criterion.weights(self.config['num_classes'], masks.device))
            losses['soft_iou_loss'] = loss              # = 1 + criterion(res, masks), reference :40
            if self.training:
                return [None] * N, losses
            out_size = tuple(masks.shape[-2:])
        logits = self._logits(run, feats, out_size, dtype)
        probs = ops.softmax2d(logits, self.config['num_classes']).permute(0, 3, 1, 2)
        return list(probs.split(counts)), losses

    def _logits(self, run, feats, out_size, dtype):
        with torch.no_grad():
            nhwc = [f.permute(0, 2, 3, 1).to(dtype) for f in feats]
            return run.forward([f if ops_is_nhwc(f) else f.contiguous() for f in nhwc], out_size=out_size)
