"""Glue between the nn.Module surface and the static HIP plans: plan cache, flat gradient store, autograd hook-in."""
import os

import torch
import torch.nn as nn

from . import _lib, ops
from .plan import Plan


def compute_dtype(owner, x):
    """Arithmetic type of a forward call: bf16 under autocast (the reference trains under amp.autocast,
    train.py:455) or for half/bf16 inputs (val_nuclei.py:116,137), else the owner's `hdy_dtype`, else fp32."""
    forced = getattr(owner, 'hdy_dtype', None)
    if forced is not None:
        return forced
    if torch.is_autocast_enabled() or x.dtype in (torch.float16, torch.bfloat16):
        return torch.bfloat16
    return torch.float32


class GradStore:
    """All parameter gradients in one flat fp32 buffer.  Kernels write this step's gradient into `cur`;
    publish() folds it into `acc`, whose views are what `param.grad` points at (so optimizers, clipping and the
    RCCL all-reduce see one contiguous tensor)."""

    def __init__(self, params, device):
        self.params = [p for p in params]
        self.offsets, n = {}, 0
        for p in self.params:
            self.offsets[id(p)] = n
            n += (p.numel() + 3) // 4 * 4          # keep every view 16-byte aligned
        self.numel = n
        self.cur = torch.zeros(n, dtype=torch.float32, device=device)
        self.acc = torch.zeros(n, dtype=torch.float32, device=device)
        self._acc_views = {id(p): self._view(self.acc, p) for p in self.params}
        self._cur_views = {id(p): self._view(self.cur, p) for p in self.params}
        self.aliased = False          # parameters' .grad are views of `cur` itself (no copy was made for them yet)

    def _view(self, flat, p):
        o = self.offsets[id(p)]
        return flat[o:o + p.numel()].view(p.shape)

    def view_of(self, p):
        return self._view(self.cur, p)

    def publish(self):
        live = [p for p in self.params if p.requires_grad]
        fresh = all(p.grad is None for p in live)
        if fresh:
            # hand out views of the buffer the backward wrote — no 29 MB copy per step.  They stay valid until the next backward pass
            # starts; if the caller still holds them then (gradient accumulation, zero_grad(set_to_none=False)), before_backward()
            # moves them into `acc` first
            for p in live:
                p.grad = self._cur_views[id(p)]
            self.aliased = True
            return
        mine = all(p.grad is not None and p.grad.data_ptr() == self._acc_views[id(p)].data_ptr() for p in live)
        if mine:
            self.acc.add_(self.cur)
            return
        for p in live:                               # mixed ownership: fall back to per-parameter accumulation
            g = self.view_of(p)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.add_(g)

    def before_backward(self):
        """Called before anything of a backward pass writes `cur`: gradients that are still views of it move to `acc`."""
        if not self.aliased:
            return
        self.aliased = False
        held = [p for p in self.params if p.grad is not None and p.grad.data_ptr() == self._cur_views[id(p)].data_ptr()]
        if held:
            self.acc.copy_(self.cur)
            for p in held:
                p.grad = self._acc_views[id(p)]


class _PlanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, engine, plan, hook, images):
        ctx.engine, ctx.plan = engine, plan
        return tuple(plan.run_forward(images))

    @staticmethod
    def backward(ctx, *gdets):
        ctx.plan.run_backward(gdets)
        ctx.engine.after_backward()
        return None, None, None, None


class _FusedLossMaskFn(torch.autograd.Function):
    """_FusedLossFn for a model with a mask branch: also returns the ordering token (see _PlanMaskFn)."""

    @staticmethod
    def forward(ctx, engine, plan, hook, loss_view):
        ctx.engine, ctx.plan = engine, plan
        return loss_view.clone(), torch.zeros(1, device=loss_view.device)

    @staticmethod
    def backward(ctx, g, _token_grad):
        ctx.plan.run_backward(None, scale=g)
        ctx.engine.after_backward()
        return None, None, None, None


class _PlanMaskFn(torch.autograd.Function):
    """_PlanFn for a model with a mask branch: besides the logits it returns a 1-element token that the mask branch's autograd
    node consumes, so that autograd runs the branch's backward (which fills the plan's mask-feature gradients) before this one."""

    @staticmethod
    def forward(ctx, engine, plan, hook, images):
        ctx.engine, ctx.plan = engine, plan
        return tuple(plan.run_forward(images)) + (torch.zeros(1, device=images.device),)

    @staticmethod
    def backward(ctx, *grads):
        ctx.plan.run_backward(grads[:-1])
        ctx.engine.after_backward()
        return None, None, None, None


class MaskBranchFn(torch.autograd.Function):
    """roi_align over the plan's mask feature maps + the Mask R-CNN head (hd_yolo_amd/maskhead.py), as one autograd node:
    forward -> fp32 logits (R, nc_masks, 28, 28); backward -> head parameter gradients into the flat store and the roi_align
    scatter into the plan's mask-feature gradient buffers."""

    @staticmethod
    def forward(ctx, token, engine, plan, head, rois_by_level, order, dtype):
        from .maskhead import MaskHeadRun
        P = head.mask_output_size // 2
        feats = plan.mask_features()
        parts = [ops.roi_align(feats[l], r, 1.0 / head._stride_cached(l), P, 2, head.aligned) for l, r in enumerate(rois_by_level)]
        x = torch.cat(parts)[order]
        run = MaskHeadRun(head.seg_h, dtype)
        logits = run.forward(x.contiguous(), train=True)
        ctx.engine, ctx.plan, ctx.head, ctx.run = engine, plan, head, run
        ctx.rois, ctx.order, ctx.sizes = rois_by_level, order, [p.shape[0] for p in parts]
        return logits.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dlogits):
        store, plan, head = ctx.engine.store, ctx.plan, ctx.head
        store.before_backward()
        dx = ctx.run.backward(dlogits.permute(0, 2, 3, 1).contiguous().float(), store.view_of)
        inv = torch.empty_like(ctx.order)
        inv[ctx.order] = torch.arange(len(ctx.order), device=ctx.order.device)
        dparts = torch.split(dx[inv], ctx.sizes)                  # back to the per-level concatenation order
        for l, (v, rois, dpart) in enumerate(zip(plan.mask_vals, ctx.rois, dparts)):
            img = ops.roi_align_bwd(dpart.contiguous(), (v.n, v.h, v.w, v.c), rois, 1.0 / head._stride_cached(l), 2, head.aligned)
            ops.cast_store(img, v.g(), accumulate=False)
        plan.mask_grads_ready = True
        return torch.zeros(1, device=dlogits.device), None, None, None, None, None, None


class _PlanTapFn(torch.autograd.Function):
    """_PlanFn for a plan with tapped layers: also returns the tapped feature maps, so that outside modules (hnet's segmentation
    header) hang off them in autograd and their gradients come back here before the backward launch list runs."""

    @staticmethod
    def forward(ctx, engine, plan, hook, images):
        ctx.engine, ctx.plan = engine, plan
        dets = tuple(plan.run_forward(images))
        ctx.ndet = len(dets)
        return dets + tuple(plan.tap_features())

    @staticmethod
    def backward(ctx, *grads):
        ctx.plan.run_backward(grads[:ctx.ndet], gtaps=grads[ctx.ndet:])
        ctx.engine.after_backward()
        return None, None, None, None


class _FusedLossTapFn(torch.autograd.Function):
    """_FusedLossFn for a plan with tapped layers (see _PlanTapFn)."""

    @staticmethod
    def forward(ctx, engine, plan, hook, loss_view):
        ctx.engine, ctx.plan = engine, plan
        return (loss_view.clone(),) + tuple(plan.tap_features())

    @staticmethod
    def backward(ctx, g, *gtaps):
        ctx.plan.run_backward(None, scale=g if g is not None else torch.zeros(1, device=gtaps[0].device), gtaps=gtaps)
        ctx.engine.after_backward()
        return None, None, None, None


class _FusedLossFn(torch.autograd.Function):
    """loss = hdy_det_loss(plan logits, targets); its backward replays the plan's backward launch list."""

    @staticmethod
    def forward(ctx, engine, plan, hook, loss_view):
        ctx.engine, ctx.plan = engine, plan
        return loss_view.clone()

    @staticmethod
    def backward(ctx, g):
        ctx.plan.run_backward(None, scale=g)
        ctx.engine.after_backward()
        return None, None, None, None


class Engine:
    """Owns the plans of one (backbone, neck, head) triple."""

    def __init__(self, backbone, neck=None, head=None, max_plans=None, extra=None, taps=()):
        self.parts = (backbone, neck, head)
        self.extra = extra            # module(s) outside the plans whose parameters share the flat gradient store (hnet's seg header)
        self.taps = tuple(taps)       # layer indices whose outputs those modules read
        self.plans = {}
        # plans kept per (shape, dtype, mode): train.py --multi-scale walks a handful of sizes, each with its own static plan (oldest evicted)
        self.max_plans = max_plans if max_plans is not None else int(os.environ.get('HDY_MAX_PLANS', '8'))
        self.store = None
        self.hook = None
        self.grad_hooks = []          # callables run after every backward, before publish (data-parallel all-reduce)
        self.bucket_hooks = []        # callables (store, a, b, side_stream) run DURING backward when flat gradient range [a, b) is final
        self.mask_token = None
        self.sync_bn = None           # SyncBatchNorm: None, True (default process group) or a group; set by parallel.DataParallel(sync_bn=...)

    # An engine is a cache of device buffers and marshalled launch records for the module OBJECTS it was traced from: a copied or
    # unpickled module (ModelEMA, Deploy(fuse=True), torch.save of a whole model) gets none and builds its own on first use.
    def __deepcopy__(self, memo):
        return None

    def __reduce__(self):
        return (type(None), ())

    def _scan(self):
        """One walk over the module trees of the parts: (parameters in registration order, their ids, requires_grad flags, which Conv
        modules still own a BatchNorm).  Done on every call (a caller may have swapped parameters or fused / frozen modules since
        the last one) — hence a bare stack walk over `_modules` / `_parameters`: nn.Module.parameters() + .modules() cost 1 ms per
        call on yolov5s, more than the whole inference launch list."""
        params, seen, fused, bufs = [], set(), [], []
        stack = []
        for part in self.parts + (self.extra,):
            if part is None:
                continue
            stack.extend(reversed(list(part) if not isinstance(part, nn.Module) else [part]))
        mods_seen = set()
        while stack:
            m = stack.pop()
            if id(m) in mods_seen:
                continue
            mods_seen.add(id(m))
            if type(m).__name__ == 'Conv':
                fused.append('bn' in m._modules)
            for q in m._parameters.values():
                if q is not None and id(q) not in seen:
                    seen.add(id(q))
                    params.append(q)
            for q in m._buffers.values():               # BatchNorm running statistics: the launch records hold their addresses too
                if q is not None and q.is_floating_point():
                    bufs.append(q.data_ptr())
            stack.extend(reversed([c for c in m._modules.values() if c is not None]))
        # identity AND storage: `model.cpu(); model.cuda()` or `.to(dtype)` round trips keep the parameter objects but move their data,
        # and the launch records / pack tables / gradient views of a plan bake raw device addresses
        return params, (tuple((id(q), q.data_ptr()) for q in params), tuple(bufs)), tuple(q.requires_grad for q in params), tuple(fused)

    def _params(self):
        return self._scan()[0]

    def _check_and_sign(self):
        """Plans and the flat gradient store hold the parameter OBJECTS they were traced with: when a caller swaps some (e.g.
        manipulate_header_label_order replaces the header's detection convs) everything is rebuilt.  Returns the plan-cache
        signature: which convs are fused, and the requires_grad flags that select the backward launches (Model.freeze)."""
        _, ids, flags, fused = self._scan()
        if ids != self.__dict__.get('_param_ids'):
            if '_param_ids' in self.__dict__:
                self.plans.clear()
                self.store = None
                self.hook = None
            self._param_ids = ids
        return hash((fused, flags))

    def plan_for(self, x, training, dtype):
        ops.require_gpu(x)
        _lib.load()
        key = (tuple(x.shape), dtype, bool(training), x.device.index, self._check_and_sign(), bool(self.sync_bn) and bool(training))
        plan = self.plans.get(key)
        if plan is None:
            if training and self.store is None:
                self.store = GradStore(self._params(), x.device)
                self.hook = torch.zeros(1, device=x.device, requires_grad=True)
            if len(self.plans) >= self.max_plans:
                self.plans.pop(next(iter(self.plans)))
            b, n, h = self.parts
            tap_params = []
            if self.extra is not None and training:
                tap_params = [q for m in (self.extra if not isinstance(self.extra, nn.Module) else [self.extra]) for q in m.parameters()]
            plan = Plan(b, n, h, tuple(x.shape), dtype, training, x.device, grad_store=self.store, taps=self.taps, tap_params=tap_params, sync=self.sync_bn)
            plan.bucket_hook = self._bucket_ready
            self.plans[key] = plan
        self.last_plan = plan
        return plan

    def _bucket_ready(self, a, b, side_stream):
        for fn in self.bucket_hooks:
            fn(self.store, a, b, side_stream)

    def plan_for_features(self, feats, dtype):
        """Eval plan of (neck, head) fed with bare feature maps {layer index: NCHW tensor} (FPN.forward / Detect.forward)."""
        _lib.load()
        shapes = {k: tuple(v.shape) for k, v in feats.items() if isinstance(k, int) and k >= 0}
        dev = next(iter(feats.values())).device
        key = ('features', tuple(sorted(shapes.items())), dtype, dev.index, self._check_and_sign())
        plan = self.plans.get(key)
        if plan is None:
            if len(self.plans) >= self.max_plans:
                self.plans.pop(next(iter(self.plans)))
            _, n, h = self.parts
            plan = self.plans[key] = Plan(None, n, h, shapes, dtype, False, dev)
        return plan

    def forward(self, x, training, dtype):
        """Returns (plan, det logits list).  With grad enabled in training mode the logits are attached to autograd."""
        plan = self.plan_for(x, training, dtype)
        self.mask_token = None
        self.tap_outputs = None
        if training and torch.is_grad_enabled():
            if plan.tap_keys:
                outs = _PlanTapFn.apply(self, plan, self.hook, x)
                dets, self.tap_outputs = outs[:len(outs) - len(plan.tap_keys)], list(outs[len(outs) - len(plan.tap_keys):])
            elif plan.mask_vals:
                *dets, self.mask_token = _PlanMaskFn.apply(self, plan, self.hook, x)
            else:
                dets = _PlanFn.apply(self, plan, self.hook, x)
        else:
            dets = plan.run_forward(x)
            if plan.tap_keys:
                self.tap_outputs = plan.tap_features()
        return plan, list(dets)

    def forward_fused_loss(self, x, dtype, head, gts, tcls):
        """Training forward + fused loss kernel.  Returns (plan, loss[1] attached to autograd, items[3])."""
        plan = self.plan_for(x, True, dtype)
        plan.run_forward(x)
        plan.fused_loss(head)(gts, tcls)
        self.mask_token = None
        self.tap_outputs = None
        if plan.tap_keys:
            loss, *self.tap_outputs = _FusedLossTapFn.apply(self, plan, self.hook, plan.loss_out[0:1])
        elif plan.mask_vals:
            loss, self.mask_token = _FusedLossMaskFn.apply(self, plan, self.hook, plan.loss_out[0:1])
        else:
            loss = _FusedLossFn.apply(self, plan, self.hook, plan.loss_out[0:1])
        return plan, loss, plan.loss_out[1:4].clone()

    def tap_grad_sink(self, plan):
        """`grad_of(param)` for an outside module's backward: the flat store's view; the first call marks the gradients as written"""
        def grad_of(p):
            self.store.before_backward()
            plan.tap_grads_ready = True
            return self.store.view_of(p)
        return grad_of

    def after_backward(self):
        for fn in self.grad_hooks:
            fn(self.store)
        self.store.publish()


class _Seq(list):
    """A bare list of modules posing as a backbone for single-module plans."""
    save = [0]


def module_forward(module, x):
    """Standalone forward of one hot-path module (Conv, Bottleneck, C3, SPPF) on an NCHW fp32 CUDA tensor: runs a
    one-module HIP plan and returns the NCHW-shaped result (forward only; training goes through Model)."""
    ops.require_gpu(x)
    eng = module.__dict__.get('_hdy_engine')
    if eng is None:
        eng = Engine(_Seq([module]))
        object.__setattr__(module, '_hdy_engine', eng)
    plan = eng.plan_for(x, False, compute_dtype(module, x))
    plan.run_forward(x)
    return plan.feature(0)
