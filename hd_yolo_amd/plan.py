"""Static execution plan for the metayolo backbone + neck + detection convs on MI355X.

The reference runs the network as ~200 eager ATen calls per forward plus autograd's dynamic graph
(metayolo/models/yolov5.py:53-59, :68-77; layers.py:37-38; train.py:457,472).  Here the module tree is traced
ONCE per (input shape, arithmetic type, train/eval) into a flat list of kernel launches over pre-allocated
NHWC buffers; forward and backward are replays of two launch lists.  What the tracing decides:

  * layout      activations NHWC bf16/fp32 in HBM, never NCHW; torch.cat along C is free: every producer writes
                straight into its channel slice of the concat buffer (pixel pitch = concat width).
  * fusion      C3's cv1 and cv2 (two 1x1 convs on the same input) become ONE conv with K = 2c_ (train mode);
                BN statistics come out of the conv kernel's epilogue as per-tile slabs; normalise+SiLU(+residual)
                is one streaming pass; in eval mode BN is folded into the conv epilogue (scale/shift/SiLU/residual).
  * backward    reverse replay: [BN+SiLU backward -> dy] -> wgrad -> dgrad, gradients of multiply-consumed tensors are
                accumulated in the dgrad epilogue (no add kernels); the Bottleneck shortcut aliases gradient storage.
  * gradients   all parameter gradients live in one flat fp32 buffer (views per parameter) so data-parallel
                all-reduce is a few large RCCL calls (hd_yolo_amd/parallel.py).

torch is used for memory (torch.empty), streams and the autograd hook-in (one custom Function around the whole
plan); every arithmetic op inside is a HIP kernel from libhdyolo_hip.so.
"""
import os

import torch
import torch.nn as nn

from . import _lib, ops

USE_GRAPHS = os.environ.get('HDY_GRAPH', '0') == '1'   # hipGraph replay of ~400-node graphs measured slower than eager on ROCm 7.2
SIDE_WGRAD = os.environ.get('HDY_SIDE_WGRAD', '1') == '1' and not USE_GRAPHS      # weight gradients on a second stream
DY_RING = int(os.environ.get('HDY_DY_RING', '4'))
SKIP_WGRAD = os.environ.get('HDY_SKIP_WGRAD') == '1'     # measurement only: no weight-gradient launches at all
# BN-backward reduce pass served by the launch that completes dz ('1': fused 1x1 backward and dgrad launches, 'fused': the former only).
# Built, parity-tested (tests/test_gpu_kernels.py, test_gpu_model.py) and NOT the default: the epilogues that would serve the statistics are
# themselves VALU-bound (exp + rcp per element on top of the store loop) — yolov5s B=64 step 14.09 ms without, 14.31 ('fused'), 14.37 ('1').
PRODUCER_STATS = os.environ.get('HDY_PRODUCER_STATS', 'fused')       # 'fused': statistics served by the fused 1x1 backward kernel only (train step 13.47 vs 13.54 ms off); '1': by dgrad epilogues too (slower); '0': off
STEM_FUSED = os.environ.get('HDY_STEM_FUSED', '1') == '1'   # the stem's weight gradient applies its unit's BatchNorm / SiLU backward itself (no dy tensor)
FUSED_1X1 = os.environ.get('HDY_FUSED_1X1', '1') == '1'     # BN-apply + wgrad + dgrad of eligible 1x1 units in one kernel (conv1x1_bwd.hip)
# probe, off: the weight re-pack on a second stream beside the input conversion (both are needed in front of the first convolution, neither depends on
# the other).  Measured round 5, one box, alternating x3: 11.813 / 11.799 / 11.793 ms per step with it, 11.673 / 11.665 / 11.649 without — the two
# cross-stream dependencies cost the main queue more than the 50-us gather they hide
PACK_SIDE = os.environ.get('HDY_PACK_SIDE', '0') == '1'
FORK_MIN_PIXELS = int(os.environ.get('HDY_FORK_MIN_PIXELS', '0'))     # probe: weight gradients of layers with fewer output pixels run inline on the main stream (no fork marker)
GRAD_BUCKET_BYTES = int(os.environ.get('HDY_GRAD_BUCKET_MB', '6')) << 20      # granularity of the "these gradients are final" marks


class Val:
    """A tensor of the traced graph: logical NHWC shape + where it lives (buffer, channel offset)."""
    __slots__ = ('n', 'h', 'w', 'c', 'name', 'buf', 'off', 'cat', 'parts', 'gbuf', 'goff', 'ginit', 'galias', 'gfinal',
                 'order', 'last_use', 'index', 'needs_grad')

    def __init__(self, n, h, w, c, name, order):
        self.n, self.h, self.w, self.c, self.name, self.order = n, h, w, c, name, order
        self.buf = self.gbuf = self.cat = self.galias = self.gfinal = None
        self.off = self.goff = 0
        self.parts = None          # for concat values: [(Val, offset)]
        self.ginit = False
        self.last_use = order
        self.index = None          # model-level node index, when this is a layer output
        self.needs_grad = True     # False: nothing trainable upstream (Model.freeze): no gradient is propagated into it

    def t(self):
        return self.buf[..., self.off:self.off + self.c]

    def g(self):
        return self.gbuf[..., self.goff:self.goff + self.c]

    def gread(self):
        return self.gfinal if self.gfinal is not None else self.g()


class ConvUnit:
    def __init__(self, mods, x, res, outs, stem):
        self.mods, self.x, self.res, self.outs, self.stem = mods, x, res, outs, stem
        c = mods[0].conv
        self.k, self.s, self.p = c.kernel_size[0], c.stride[0], c.padding[0]
        self.Ks = [m.conv.out_channels for m in mods]
        self.K = sum(self.Ks)
        self.C = c.in_channels
        self.has_bn = hasattr(mods[0], 'bn')
        # torchvision-style FrozenBatchNorm2d put in place by Model.freeze (utils_torch.freeze_bn): constant scale / shift in training too
        self.frozen = self.has_bn and type(mods[0].bn).__name__ == 'FrozenBatchNorm2d'
        assert all((hasattr(m, 'bn') and type(m.bn).__name__ == 'FrozenBatchNorm2d') == self.frozen for m in mods)
        self.act = act_code(mods[0].act)


class DetUnit:
    def __init__(self, conv, x, level):
        self.conv, self.x, self.level = conv, x, level
        self.K = conv.out_channels
        self.Kp = (self.K + 7) // 8 * 8


class PoolUnit:
    def __init__(self, x, outs):
        self.x, self.outs = x, outs


class UpUnit:
    def __init__(self, x, out):
        self.x, self.out = x, out


def act_code(act):
    if isinstance(act, nn.SiLU):
        return ops.ACT_SILU
    if isinstance(act, nn.Identity):
        return ops.ACT_NONE
    raise _lib.HdyError(f'activation {type(act).__name__} has no HIP kernel on this path (SiLU / Identity only)')


def _is(m, name):
    return type(m).__name__ == name


class Plan:
    def __init__(self, backbone, neck, head, shape, dtype, training, device, grad_store=None, taps=(), tap_params=(), sync=None):
        """backbone / neck: the metayolo CSPDarkNet / FPN containers (neck, head may be None);
        head: Detect module or None; shape = (B, 3, H, W).
        Feature-input plans (FPN.forward / Detect.forward called on bare feature maps): backbone is None and shape is
        {layer index: (B, C, H, W)}; eval only."""
        self.dtype, self.training, self.device = dtype, training, device
        # SyncBatchNorm (train.py --sync-bn; reference: train.py:281-283): None = per-rank statistics; True = the default process group, or a
        # group object: every training BatchNorm all-reduces its (SUM, SUM2, count) forward and its (SUM du, SUM du*xhat, count) backward
        self.sync = sync if training else None
        self.ext_shapes = dict(shape) if isinstance(shape, dict) else None
        if self.ext_shapes is not None:
            if backbone is not None or training:
                raise _lib.HdyError('feature-input plans are neck/head-only and forward-only')
            self.B, self.Cin, self.H, self.W = next(iter(self.ext_shapes.values()))[0], 0, 0, 0
        else:
            self.B, self.Cin, self.H, self.W = shape
        self.input = None
        self.ext = {}
        self.vals, self.units = [], []
        self.grad_store = grad_store
        # taps: layer indices whose outputs also feed modules outside the plan (hnet's segmentation header on the pyramid): their
        # gradient buffers are pre-filled from outside before the backward list runs; tap_params: those modules' parameters (their
        # gradients are written into the flat store by the outside backward, or zeroed when it did not run)
        self.tap_keys, self.tap_params = list(taps), list(tap_params)
        self.tap_grads_ready = False
        self._grad_log = None
        self.bucket_hook = None         # engine: called as hook(a, b, side_stream) when gradient elements [a, b) of the flat buffer are final
        self._order = 0
        self._trace(backbone, neck, head)
        self._allocate()
        self.packs = ops.PackTable(device)          # every weight re-pack of the plan, one launch per step
        self.bn_eval = ops.BnEvalTable(device)      # eval plans: every BatchNorm's folded scale / shift, one launch per forward
        self.fwd = self._compile_forward()
        self.bwd = self._compile_backward() if training else None
        # the two launch lists have fixed pointers and shapes: after one eager run each they are captured into hipGraphs and
        # replayed with one call (the eager lists are ~350 / ~450 launches per step for yolov5s)
        self._graphs = {'fwd': None, 'bwd': None}
        self._runs = {'fwd': 0, 'bwd': 0}

    # ------------------------------------------------------------------ tracing
    def _val(self, n, h, w, c, name):
        v = Val(n, h, w, c, name, self._order)
        self._order += 1
        self.vals.append(v)
        return v

    def _use(self, v):
        v.last_use = self._order

    def _conv(self, mods, x, res=None, stem=False):
        c0 = mods[0].conv
        for m in mods:
            c = m.conv
            if c.groups != 1 or c.dilation[0] != 1 or c.kernel_size[0] != c.kernel_size[1] or c.stride[0] != c.stride[1]:
                raise _lib.HdyError('grouped / dilated / non-square convolutions are outside the hot path')
            assert (c.kernel_size, c.stride, c.padding, c.in_channels) == (c0.kernel_size, c0.stride, c0.padding, c0.in_channels)
        k, s, p = c0.kernel_size[0], c0.stride[0], c0.padding[0]
        if stem:
            n, h, w = self.B, self.H, self.W
        else:
            n, h, w = x.n, x.h, x.w
            self._use(x)
        if res is not None:
            self._use(res)
        ho, wo = ops.out_dim(h, k, s, p), ops.out_dim(w, k, s, p)
        outs = [self._val(n, ho, wo, m.conv.out_channels, 'conv') for m in mods]
        u = ConvUnit(mods, x, res, outs, stem)
        self.units.append(u)
        if res is not None and self.training:
            if res.cat is not None or res.parts is not None:
                raise _lib.HdyError('shortcut from a concat member is not plannable')
            res.galias = outs[0]
        return outs

    def _concat(self, vals):
        n, h, w = vals[0].n, vals[0].h, vals[0].w
        cat = self._val(n, h, w, sum(v.c for v in vals), 'cat')
        off, cat.parts = 0, []
        for v in vals:
            if v.cat is not None or v.parts is not None:
                raise _lib.HdyError('a tensor may sit in one concat only (nested / shared concats are not planned)')
            assert (v.n, v.h, v.w) == (n, h, w)
            v.cat, v.off = cat, off
            cat.parts.append((v, off))
            off += v.c
            self._use(v)
        return cat

    def _module(self, m, x):
        t = type(m).__name__
        if t == 'Conv':
            return self._conv([m], x, stem=(x is None))[0]
        if t == 'Bottleneck':
            h = self._conv([m.cv1], x)[0]
            return self._conv([m.cv2], h, res=x if m.add else None)[0]
        if t == 'C3':
            fuse = (self.training and hasattr(m.cv1, 'bn') and hasattr(m.cv2, 'bn') and type(m.cv1.bn) is type(m.cv2.bn)
                    and m.cv1.conv.weight.requires_grad == m.cv2.conv.weight.requires_grad)
            if fuse:
                a, b = self._conv([m.cv1, m.cv2], x)
            else:
                a, b = self._conv([m.cv1], x)[0], self._conv([m.cv2], x)[0]
            for bt in m.m:
                a = self._module(bt, a)
            return self._conv([m.cv3], self._concat([a, b]))[0]
        if t == 'SPPF':
            if m.m.kernel_size != 5:
                raise _lib.HdyError('SPPF pooling kernel must be 5')
            a = self._conv([m.cv1], x)[0]
            self._use(a)
            ys = [self._val(a.n, a.h, a.w, a.c, 'pool') for _ in range(3)]
            self.units.append(PoolUnit(a, ys))
            return self._conv([m.cv2], self._concat([a] + ys))[0]
        if t == 'Upsample':
            if m.mode != 'nearest' or float(m.scale_factor) != 2.0:
                raise _lib.HdyError('only 2x nearest upsampling is on the hot path')
            self._use(x)
            out = self._val(x.n, 2 * x.h, 2 * x.w, x.c, 'up')
            self.units.append(UpUnit(x, out))
            return out
        if t == 'Concat':
            if m.d != 1:
                raise _lib.HdyError('Concat along a dimension other than channels')
            return self._concat(x)
        if t == 'Sequential':
            for sub in m:
                x = self._module(sub, x)
            return x
        raise _lib.HdyError(f'module {t} is outside the metayolo hot path (no HIP plan for it)')

    def _trace(self, backbone, neck, head):
        outs = {}
        if backbone is None:
            for k, (b, c, h, w) in self.ext_shapes.items():
                if c % 8:
                    raise _lib.HdyError(f'feature map {k} has {c} channels: HIP plans take multiples of 8')
                outs[k] = self._val(b, h, w, c, 'input')
                outs[k].index = k
            self.ext = dict(outs)
            backbone = []
        first = backbone[0] if len(backbone) else None
        is_stem = (type(first).__name__ == 'Conv' and self.Cin == 3 and first.conv.kernel_size == (6, 6)
                   and first.conv.stride == (2, 2) and first.conv.padding == (2, 2))
        if is_stem:
            x = None                       # the 6x6/s2 stem reads the padded 4-channel image directly
        elif self.ext:
            x = None
        else:
            if self.Cin % 8:
                raise _lib.HdyError(f'input with {self.Cin} channels: only the 6x6/s2 RGB stem or channel counts that are '
                                    'multiples of 8 can enter a HIP plan')
            x = self.input = self._val(self.B, self.H, self.W, self.Cin, 'input')
        for i, m in enumerate(backbone):
            x = self._module(m, x)
            x.index = i
            outs[i] = x
        self.feature_keys = list(getattr(backbone, 'save', [len(backbone) - 1])) if len(backbone) else list(self.ext)
        if neck is not None:
            cur = None
            for m in neck:
                f = m.f
                if isinstance(f, int):
                    src = cur if f == -1 else outs[f]
                else:
                    src = [cur if j == -1 else outs[j] for j in f]
                cur = self._module(m, src)
                cur.index = m.i
                outs[m.i] = cur
            self.feature_keys = list(neck.save)
        self.outs = outs
        # one or several Detect headers on the same feature maps (yolo.py:62-81 loops over self.headers): their 1x1 detection convs are
        # all units of this plan; det_views() lists the logits header after header, level after level
        heads = [] if head is None else (list(head) if isinstance(head, (list, tuple)) else [head])
        self.det_units, self.det_split = [], []
        for hi, hd in enumerate(heads):
            f = hd.f if isinstance(hd.f, (list, tuple)) else [hd.f]
            for l, (j, conv) in enumerate(zip(f, hd.m)):
                self._use(outs[j])
                u = DetUnit(conv, outs[j], l)
                u.head, u.na, u.no = hi, hd.na, hd.no
                self.units.append(u)
                self.det_units.append(u)
            self.det_split.append(len(f))
        if heads:
            self.na, self.no = heads[0].na, heads[0].no
        if len(heads) > 1 and any(getattr(hd, 'seg', None) is not None for hd in heads):
            raise _lib.HdyError('a mask branch is supported on single-header models only')
        head = heads[0] if len(heads) == 1 else None       # the mask branch below belongs to a lone header
        # mask branch: one 3x3 Conv per level, top-down module order (yolo_head.py:123-124, :170-173); their outputs
        # feed roi_align outside the plan and their output gradients arrive from there (MaskBranch in engine.py)
        self.mask_vals = []
        if head is not None and getattr(head, 'seg', None) is not None:
            f = head.f if isinstance(head.f, (list, tuple)) else [head.f]
            vals = [self._module(m, outs[f[-i]]) for i, m in enumerate(head.seg, 1)]
            self.mask_vals = vals[::-1]
        self.mask_grads_ready = False
        self.seg_h_params = list(head.seg_h.parameters()) if self.mask_vals else []

    def _fusable_1x1(self, u):
        """Backward of this unit as ONE launch after the statistics pass (hdy_conv1x1_bwd_fused): 1x1 / stride 1 Conv + live BatchNorm +
        SiLU in bf16 with a kernel instance for its widths."""
        return (FUSED_1X1 and not USE_GRAPHS and self.training and self.dtype == torch.bfloat16 and not u.stem and u.k == 1 and u.s == 1
                and u.p == 0 and u.has_bn and not u.frozen and u.act == ops.ACT_SILU and ops.fused_1x1_ok(u.C, u.K, self.dtype)
                and all(m.conv.out_channels % 8 == 0 for m in u.mods))

    # ------------------------------------------------------------------ memory
    def _new(self, *shape, dtype=None, zero=False):
        f = torch.zeros if zero else torch.empty
        return f(shape, dtype=dtype or self.dtype, device=self.device)

    def _allocate(self):
        dt = self.dtype
        for v in self.vals:
            if v.parts is not None or v.cat is None:
                v.buf = self._new(v.n, v.h, v.w, v.c)
        for v in self.vals:
            if v.cat is not None:
                v.buf = v.cat.buf
        self.prep = self._new(self.B, self.H + 4, self.W + 4, 4)
        f32 = torch.float32
        max_stats = max_dy = max_wg = max_bnws = max_f1 = 1
        for u in self.units:
            if isinstance(u, ConvUnit):
                o = u.outs[0]
                M = o.n * o.h * o.w
                kind = ops.PACK_STEM if u.stem else ops.PACK_FWD
                if self.training or len(u.mods) > 1:
                    u.wp = ops.pack_alloc(u.K, u.C, u.k, u.k, u.s, u.p, kind, dt, self.device)
                else:
                    u.wp = ops.pack_alloc(u.K, u.C, u.k, u.k, u.s, u.p, kind, dt, self.device)
                u.scale, u.shift = self._new(u.K, dtype=f32), self._new(u.K, dtype=f32)
                if self.training:
                    if not u.has_bn:
                        raise _lib.HdyError('training a fused (BN-folded) model is not supported: build the model unfused')
                    u.yraw = self._new(o.n, o.h, o.w, u.K)
                    hin, win = (self.H, self.W) if u.stem else (u.x.h, u.x.w)
                    if u.frozen:
                        u.mean = u.invstd = None
                        u.mtiles = 0
                    else:
                        u.mean, u.invstd = self._new(u.K, dtype=f32), self._new(u.K, dtype=f32)
                        u.mtiles = ops.stat_slabs(o.n, hin, win, u.C, u.K, u.k, u.k, u.s, u.p, dt)
                    max_stats = max(max_stats, u.mtiles * 2 * u.K)
                    max_dy = max(max_dy, M * u.K)
                    wgb = ops.wgrad_ws_bytes(o.n, hin, win, u.C, u.K, u.k, u.k, u.s, u.p, dt, stem=u.stem)
                    max_wg = max(max_wg, wgb)
                    max_bnws = max(max_bnws, ops.bn_bwd_ws_floats(M, u.K))
                    if self._fusable_1x1(u):
                        max_f1 = max(max_f1, ops.fused_1x1_ws_bytes(M, u.C, u.K))
                        wgb = ops.fused_1x1_ws_bytes(M, u.C, u.K)
                    # batched split reductions: the slabs of a unit stay untouched until its bucket's one reduction launch has read them
                    if not u.stem:
                        u.wpd = ops.pack_alloc(u.K, u.C, u.k, u.k, u.s, u.p, ops.PACK_DGRAD, dt, self.device)
            elif isinstance(u, DetUnit):
                x = u.x
                u.wp = ops.pack_alloc(u.K, x.c, 1, 1, 1, 0, ops.PACK_FWD, dt, self.device)
                u.logits = self._new(x.n, x.h, x.w, u.Kp, dtype=f32, zero=True)
                if self.training:
                    M = x.n * x.h * x.w
                    u.wpd = ops.pack_alloc(u.Kp, x.c, 1, 1, 1, 0, ops.PACK_DGRAD, dt, self.device)
                    wgb = ops.wgrad_ws_bytes(x.n, x.h, x.w, x.c, u.Kp, 1, 1, 1, 0, dt)
                    max_wg = max(max_wg, wgb)
                    max_bnws = max(max_bnws, ops.bn_bwd_ws_floats(M, u.Kp))
            elif isinstance(u, PoolUnit):
                if self.training:
                    a = u.x
                    u.idx = [self._new(a.n, a.h, a.w, a.c, dtype=torch.uint8) for _ in range(3)]
        if not self.training:
            return
        # logits gradients of all levels live in one flat buffer (one scale launch, one owner)
        sizes = [u.x.n * u.x.h * u.x.w * u.Kp for u in self.det_units]
        self.gdet_flat = self._new(max(sum(sizes), 1), zero=True)
        off = 0
        for u, n in zip(self.det_units, sizes):
            u.gdet = self.gdet_flat[off:off + n].view(u.x.n, u.x.h, u.x.w, u.Kp)
            off += n
        self.loss_out = self._new(4, dtype=f32, zero=True)
        self.loss_call = None
        self.stats = self._new(max_stats, dtype=f32)
        # BN-backward output of the layer in flight; a small ring, so that the weight-gradient kernels of the previous layers
        # (side stream) may still be reading theirs while the main stream moves on
        self.dy_ring = [self._new(max_dy) for _ in range(DY_RING if SIDE_WGRAD else 1)]
        self.dy = self.dy_ring[0]
        self.wg_ws = self._new(max_wg // 4 + 16, dtype=f32)
        self.wg_ws_main = self._new(max_wg // 4 + 16, dtype=f32) if (FORK_MIN_PIXELS > 0 and SIDE_WGRAD) else self.wg_ws
        self.bn_c12 = self._new(2, kmax if False else max(u.K for u in self.units if isinstance(u, ConvUnit)), dtype=f32)     # c1 / c2 of the unit in flight
        self.f1_ws = self._new(max_f1 // 4 + 16, dtype=f32)      # weight-gradient slabs of the fused 1x1 backward (main stream: not shared with wg_ws)
        self.bn_ws = self._new(max_bnws, dtype=f32)
        kmax = max(u.K for u in self.units if isinstance(u, ConvUnit))
        self.fin_ws = self._new(32 * 2 * kmax, dtype=torch.float64)
        self.sync_sums = self._new(2 * kmax + 1, dtype=torch.float64) if self.sync else None
        # gradient storage mirrors activation storage
        for v in self.vals:
            if v.parts is not None or v.cat is None:
                v.gbuf = None       # allocated below unless aliased
        for v in reversed(self.vals):
            if v.cat is not None:
                continue
            if v.galias is not None and v.parts is None:
                continue
            v.gbuf = self._new(v.n, v.h, v.w, v.c)
        for v in self.vals:
            if v.cat is not None:
                v.gbuf, v.goff = v.cat.gbuf, v.off
        for v in reversed(self.vals):      # shortcut inputs share the gradient storage of the block output
            if v.galias is not None and v.cat is None:
                o = v.galias
                v.gbuf, v.goff = o.gbuf, o.goff
        for u in self.units:
            if isinstance(u, PoolUnit):
                u.x.gfinal = self._new(u.x.n, u.x.h, u.x.w, u.x.c)

    # ------------------------------------------------------------------ parameter access
    def _bn(self, m):
        return m.bn.weight, m.bn.bias, m.bn.running_mean, m.bn.running_var

    def _grad_views(self, p):
        # compiling the backward list: remember which launch record produces this parameter's gradient (the one appended next)
        if self._grad_log is not None:
            self._grad_log.append((p, self._grad_pos()))
        return self.grad_store.view_of(p)

    # ------------------------------------------------------------------ forward
    def _compile_forward(self):
        recs = []
        t = self.training
        for u in self.units:
            if isinstance(u, ConvUnit):
                x = self.prep if u.stem else u.x.t()
                stem_hw = (self.H, self.W) if u.stem else None
                kind = ops.PACK_STEM if u.stem else ops.PACK_FWD
                o0 = u.outs[0]
                if t:
                    wb = u.mods[1].conv.weight if len(u.mods) > 1 else None
                    self.packs.add(u.mods[0].conv.weight, wb, u.s, u.p, kind, u.wp)
                    M = o0.n * o0.h * o0.w
                    stats = None if u.frozen else self.stats[:u.mtiles * 2 * u.K].view(u.mtiles, 2, u.K)
                    recs.append(ops.rec_conv_fwd(x, u.wp, u.yraw, u.K, u.k, u.k, u.s, u.p, stats=stats, stem_hw=stem_hw))
                    if self.sync and not u.frozen:
                        # SyncBatchNorm: slabs -> [SUM | SUM2 | count] in fp64, all-reduced, then the usual finalize from the global sums
                        buf = self.sync_sums[:2 * u.K + 1]
                        recs.append(ops.rec_bn_slab_sums(stats, u.mtiles, u.K, M, buf))
                        recs.append(self._sync_call(buf))
                        pair_fwd = len(u.mods) == 2 and u.res is None
                        k0 = 0
                        for i, (m, o) in enumerate(zip(u.mods, u.outs)):
                            K = m.conv.out_channels
                            recs.append(ops.rec_bn_finalize_sums(buf, u.K, k0, K, K, self._bn(m), None, u.scale[k0:], u.shift[k0:], u.mean[k0:], u.invstd[k0:],
                                                                 eps=m.bn.eps, momentum=m.bn.momentum))
                            if not pair_fwd:
                                res = u.res.t() if u.res is not None else None
                                recs.append(ops.rec_bn_act_fwd(u.yraw[..., k0:k0 + K], u.scale[k0:k0 + K], u.shift[k0:k0 + K], o.t(), res=res, act=u.act))
                            k0 += K
                        if pair_fwd:
                            recs.append(ops.rec_bn_act_fwd_pair(u.yraw, u.scale, u.shift, u.outs[0].t(), u.outs[1].t(), act=u.act))
                        continue
                    if len(u.mods) == 2 and not u.frozen and u.res is None:
                        # the C3 pair: per-channel BatchNorm over the whole 2c-wide raw tensor in one finalize + one apply pass
                        Ka = u.mods[0].conv.out_channels
                        assert u.mods[0].bn.eps == u.mods[1].bn.eps and u.mods[0].bn.momentum == u.mods[1].bn.momentum
                        recs.append(ops.rec_bn_finalize_pair(stats, u.mtiles, u.K, Ka, M, self._bn(u.mods[0]), self._bn(u.mods[1]), u.scale, u.shift,
                                                             u.mean, u.invstd, eps=u.mods[0].bn.eps, momentum=u.mods[0].bn.momentum, ws=self.fin_ws))
                        recs.append(ops.rec_bn_act_fwd_pair(u.yraw, u.scale, u.shift, u.outs[0].t(), u.outs[1].t(), act=u.act))
                        continue
                    k0 = 0
                    for m, o in zip(u.mods, u.outs):
                        K = m.conv.out_channels
                        g, b, rm, rv = self._bn(m)
                        if u.frozen:
                            recs.append(ops.rec_bn_eval_coeffs(g, b, rm, rv, u.scale[k0:k0 + K], u.shift[k0:k0 + K], eps=m.bn.eps))
                        else:
                            recs.append(ops.rec_bn_finalize(stats[:, :, k0:], u.mtiles, K, M, g, b, rm, rv, u.scale[k0:], u.shift[k0:],
                                                            u.mean[k0:], u.invstd[k0:], stats_ld=u.K, ws=self.fin_ws))
                        res = u.res.t() if u.res is not None else None
                        recs.append(ops.rec_bn_act_fwd(u.yraw[..., k0:k0 + K], u.scale[k0:k0 + K], u.shift[k0:k0 + K], o.t(), res=res, act=u.act))
                        k0 += K
                else:
                    m, o = u.mods[0], u.outs[0]
                    self.packs.add(m.conv.weight, None, u.s, u.p, kind, u.wp)
                    if u.has_bn:
                        g, b, rm, rv = self._bn(m)
                        self.bn_eval.add(g, b, rm, rv, u.scale, u.shift, eps=m.bn.eps)
                        scale, shift = u.scale, u.shift
                    else:
                        scale, shift = None, m.conv.bias
                    res = u.res.t() if u.res is not None else None
                    recs.append(ops.rec_conv_fwd(x, u.wp, o.t(), u.K, u.k, u.k, u.s, u.p, scale=scale, shift=shift, act=u.act, stem_hw=stem_hw,
                                                 res=res))
            elif isinstance(u, PoolUnit):
                idx = u.idx if t else None
                recs.append(ops.rec_sppf_pool_fwd(u.x.t(), u.outs[0].t(), u.outs[1].t(), u.outs[2].t(), idx))
            elif isinstance(u, UpUnit):
                recs.append(ops.rec_upsample_fwd(u.x.t(), u.out.t()))
            elif isinstance(u, DetUnit):
                self.packs.add(u.conv.weight, None, 1, 0, ops.PACK_FWD, u.wp)
                recs.append(ops.rec_conv_fwd(u.x.t(), u.wp, u.logits[..., :u.K], u.K, 1, 1, 1, 0, shift=u.conv.bias))
        return recs

    def _replay(self, key, recs):
        if not USE_GRAPHS:
            if not ops.USE_EXEC:
                return ops.run(recs)
            # the list is static: compiled once into words for hdy_exec_run (one C call per stretch between host callbacks) and replayed
            progs = self.__dict__.setdefault('_progs', {})
            prog = progs.get(key)
            if prog is None or prog.records is not recs or prog.nrec != len(recs):
                prog = progs[key] = ops.Program(recs)
            return prog.run()
        g = self._graphs[key]
        if g is None:
            self._runs[key] += 1
            if self._runs[key] < 2:
                return ops.run(recs)                 # first call eager: lazy one-time host setup happens here
            g = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(g):
                ops.run(recs)
            self._graphs[key] = g
        g.replay()

    def bn_counters(self):
        return [m.bn.num_batches_tracked for u in self.units if isinstance(u, ConvUnit) and u.has_bn and not u.frozen for m in u.mods]

    def bn_running(self):
        """running_mean / running_var of every live BatchNorm: the training forward writes them through raw pointers"""
        if getattr(self, '_bn_running', None) is None:
            self._bn_running = [t for u in self.units if isinstance(u, ConvUnit) and u.has_bn and not u.frozen for m in u.mods
                                for t in (m.bn.running_mean, m.bn.running_var)]
        return self._bn_running

    def run_forward_features(self, feats):
        """Feature-input plan: feats {layer index: NCHW tensor} -> det logits views (empty without a head)."""
        for k, v in self.ext.items():
            f = feats[k]
            ops.require_gpu(f)
            assert tuple(f.shape) == (v.n, v.c, v.h, v.w), (k, tuple(f.shape), (v.n, v.c, v.h, v.w))
            ops.run([ops.rec_nchw_to_nhwc(f.float().contiguous(), v.t())])
        self.packs.run(skip_unchanged=not self.training)
        self.bn_eval.run(skip_unchanged=not self.training)
        self._replay('fwd', self.fwd)
        return self.det_views()

    def run_forward(self, images):
        ops.require_gpu(images)
        if images.dtype != torch.float32 or not images.is_contiguous():
            images = images.float().contiguous()
        assert tuple(images.shape) == (self.B, self.Cin, self.H, self.W), (images.shape, self.B, self.H, self.W)
        assert images.shape[1] == self.Cin
        pack_side = self.training and PACK_SIDE
        if pack_side:
            # the per-step weight re-pack (one launch, ~50 us of gathers at yolov5s) depends on the optimizer's update, not on the images: it runs on a
            # second stream beside the input conversion (an HBM stream of ~90 us) and joins in front of the first convolution
            main = torch.cuda.current_stream(self.device)
            if self.__dict__.get('_pack_stream') is None:
                self._pack_stream = torch.cuda.Stream(device=self.device)
            self._pack_stream.wait_stream(main)
            with torch.cuda.stream(self._pack_stream):
                self.packs.run(skip_unchanged=False)
        if self.input is None:
            ops.run([ops.rec_stem_prep(images, self.prep)])
        else:
            ops.run([ops.rec_nchw_to_nhwc(images, self.input.t())])
        if pack_side:
            main.wait_stream(self._pack_stream)
        else:
            self.packs.run(skip_unchanged=not self.training)
        self.bn_eval.run(skip_unchanged=not self.training)
        self._replay('fwd', self.fwd)
        if self.training:
            torch._foreach_add_(self.bn_counters(), 1)
            # the kernels updated the running statistics in place: move their version counters as torch's own BatchNorm would, so that
            # `_version`-keyed caches (ops.BnEvalTable of an eval plan on the same live model) see the change
            torch._C._increment_version(self.bn_running())
        return self.det_views()

    def det_views(self):
        views = []
        for u in self.det_units:
            x = u.x
            views.append(u.logits[..., :u.K].view(x.n, x.h, x.w, u.na, u.no).permute(0, 3, 1, 2, 4))
        return views

    def mask_features(self):
        """NHWC views of the mask branch's per-level feature maps (level order)."""
        return [v.t() for v in self.mask_vals]

    def feature(self, index):
        """NCHW-shaped (channels-last strided) view of layer `index`'s output."""
        return self.outs[index].t().permute(0, 3, 1, 2)

    # ------------------------------------------------------------------ backward
    def _contrib(self, v):
        """Is a gradient contribution to v an accumulation?  Marks the storage initialised."""
        if v.cat is not None:
            if v.cat.ginit or v.ginit:
                return True
            v.ginit = True
            return False
        if v.galias is not None:
            return True            # storage already holds the block output's gradient (shortcut identity)
        acc = v.ginit
        v.ginit = True
        return acc

    def _sync_call(self, buf):
        """launch-list record that all-reduces a SyncBatchNorm sums buffer in place (on the current stream, as every other record)"""
        import torch.distributed as dist
        group = None if self.sync is True else self.sync
        return ('@call', lambda buf=buf, group=group: dist.all_reduce(buf, group=group))

    def _compile_backward(self):
        recs = []
        self._grad_log = []
        self._grad_pos = lambda: len(recs)
        for v in self.vals:
            v.ginit = False
        # Weight gradients are consumed only by the optimizer: with SIDE_WGRAD they run on a second stream beside the
        # dgrad / BN-backward chain (which is what the next layer waits for), filling the CUs that the many small launches
        # of the 20x20 and 40x40 layers leave idle.
        side = ops.SideStream(self.device) if SIDE_WGRAD else None
        nfork = [0]

        def wgrad(rec, reads_dy_slot=None, pixels=None):
            if SKIP_WGRAD:          # timing experiment only (gradients wrong): what the step costs without the weight-gradient stream
                return
            if callable(rec):       # rec(workspace): an inline launch must not share the side stream's split-slab workspace
                inline = side is None or (pixels is not None and pixels < FORK_MIN_PIXELS)
                rec = rec(self.wg_ws_main if (inline and side is not None) else self.wg_ws)
                if inline:
                    recs.append(rec)
                    return
            if side is None:
                recs.append(rec)
                return
            recs.append(('@fork', side, [rec], nfork[0]))
            if reads_dy_slot is not None:
                slot_user[reads_dy_slot] = nfork[0]
            nfork[0] += 1

        slot_user = {}
        nconv = 0
        # Producer-side statistics: `last[id(v)]` = the launch record that wrote the LAST contribution of v's gradient, when that is a
        # data-gradient launch which can serve statistics (index into recs, channel offset of v inside the producer's output, how to
        # rebuild the record with requests, slab count).  A unit whose outputs all have such a producer skips its reduce pass: the
        # producers' epilogues leave (SUM du, SUM du*xhat) slabs and a finalize launch turns them into dgamma / dbeta / c1 / c2.
        last, pending, remake = {}, {}, {}

        def note_grad(xv, kind, idx=None, make=None, slabs=0):
            for pv, off in ([(xv, 0)] if xv.parts is None else xv.parts):
                last[id(pv)] = (kind, idx, off, slabs)
            if idx is not None:
                remake[idx] = make

        # Model.freeze: a tensor needs a gradient only if something trainable lies upstream of it; units without trainable
        # parameters below frozen inputs are skipped altogether, frozen filters skip their weight gradient
        def trainable(u):
            if isinstance(u, DetUnit):
                return u.conv.weight.requires_grad or u.conv.bias.requires_grad
            ps = [m.conv.weight for m in u.mods]
            if u.has_bn and not u.frozen:
                ps += [q for m in u.mods for q in (m.bn.weight, m.bn.bias)]
            return any(q.requires_grad for q in ps)

        if self.input is not None:
            self.input.needs_grad = False
        for v in self.ext.values():
            v.needs_grad = False
        for k in self.tap_keys:
            v = self.outs[k]
            if v.cat is not None or v.parts is not None or v.galias is not None:
                raise _lib.HdyError(f'layer {k} cannot be tapped: its gradient storage is shared (concat member / shortcut)')
            v.ginit = True                  # holds the outside consumer's gradient when the list starts: everything else accumulates

        def up(v):                                   # needs_grad of a (possibly concatenated) input value
            if v is None:
                return False
            if v.parts is not None:
                v.needs_grad = any(pv.needs_grad for pv, _ in v.parts)
            return v.needs_grad

        for u in self.units:
            if isinstance(u, ConvUnit):
                ng = trainable(u) or up(u.x) or up(u.res)
                for o in u.outs:
                    o.needs_grad = ng
            elif isinstance(u, PoolUnit):
                for o in u.outs:
                    o.needs_grad = up(u.x)
            elif isinstance(u, UpUnit):
                u.out.needs_grad = up(u.x)
        first_conv = next((q for q in self.units if isinstance(q, ConvUnit)), None)      # = the last unit the reversed walk reaches
        for u in reversed(self.units):
            if isinstance(u, DetUnit):
                x = u.x
                tmp = self._det_bias_tmp(u)
                recs.append(ops.rec_colsum(u.gdet, tmp, self.bn_ws))
                gb = self._grad_views(u.conv.bias)
                recs.append(ops.rec_copy_f32(tmp[:u.K], gb))                                    # Kp-padded column sums -> the bias gradient
                gw = self._grad_views(u.conv.weight)
                wgrad(lambda ws, x=x, u=u, gw=gw: ops.rec_conv_wgrad(x.t(), u.gdet, gw, None, 1, 1, 1, 0, ws), pixels=x.n * x.h * x.w)
                if not up(x):
                    continue
                self.packs.add(u.conv.weight, None, 1, 0, ops.PACK_DGRAD, u.wpd, K=u.Kp)
                acc_x = self._contrib(x)
                mk = (lambda st, u=u, x=x, acc_x=acc_x: ops.rec_conv_dgrad(u.gdet, u.wpd, x.g(), 1, 1, 1, 0, accumulate=acc_x, stats=st))
                recs.append(mk(None))
                note_grad(x, 'dgrad', len(recs) - 1, mk, ops.conv_dgrad_stat_slabs(x.n, x.h, x.w, x.c, u.Kp, 1, 1, 1, 0, self.dtype) if self.dtype == torch.bfloat16 else 0)
            elif isinstance(u, UpUnit):
                if up(u.x):
                    recs.append(ops.rec_upsample_bwd(u.out.gread(), u.x.g(), accumulate=self._contrib(u.x)))
                    note_grad(u.x, 'other')
            elif isinstance(u, PoolUnit):
                a = u.x
                if not up(a):
                    continue
                gs = [a.g()] + [o.g() for o in u.outs]
                recs.append(ops.rec_sppf_pool_bwd(gs[0], gs[1], gs[2], gs[3], u.idx, a.gfinal))
                note_grad(a, 'other')
            elif isinstance(u, ConvUnit):
                if not u.outs[0].needs_grad:
                    continue
                o0 = u.outs[0]
                sync = bool(self.sync) and u.has_bn and not u.frozen
                fused = self._fusable_1x1(u) and not sync
                pair = len(u.mods) == 2 and not u.frozen
                # the stem has no data gradient: its dy has one reader, the weight-gradient kernel, which can apply the BatchNorm / SiLU backward
                # itself while it stages the tile (no apply pass, no dy tensor).  It is the last unit of the backward list, so the c1 / c2 it
                # reads from the statistics workspace on the side stream are not overwritten before the list's final join.
                stem_fused = (STEM_FUSED and u.stem and not sync and not u.frozen and u.has_bn and len(u.mods) == 1 and u is first_conv and
                              u.mods[0].conv.weight.requires_grad and ops.wgrad_stem_fused_ok(self.B, self.H, self.W, u.K, self.dtype))
                dy = None
                if not fused and not stem_fused:
                    slot = nconv % len(self.dy_ring)
                    nconv += 1
                    if slot in slot_user:              # the weight gradient that last read this ring slot must be done
                        recs.append(('@join', side, slot_user.pop(slot)))
                    dy = self.dy_ring[slot][:o0.n * o0.h * o0.w * u.K].view(o0.n, o0.h, o0.w, u.K)
                # BatchNorm / SiLU backward: statistics (reduce + finalize) and, unless the fused kernel applies them, dy.
                M = o0.n * o0.h * o0.w
                c1, c2 = self.bn_c12[0, :u.K], self.bn_c12[1, :u.K]
                prod = None
                if sync:
                    # SyncBatchNorm: local statistics pass (local dgamma / dbeta: the gradient all-reduce sums them), its partial slabs -> fp64 sums
                    # -> all-reduce -> c1 / c2 of the GLOBAL batch -> one apply pass
                    nb = ops.bn_bwd_blocks(M)
                    k0 = 0
                    for m, o in zip(u.mods, u.outs):
                        K = m.conv.out_channels
                        buf = self.sync_sums[:2 * K + 1]
                        recs.append(ops.rec_bn_act_bwd(o.gread(), u.yraw[..., k0:k0 + K], u.scale[k0:k0 + K], u.shift[k0:k0 + K], u.mean[k0:k0 + K],
                                                       u.invstd[k0:k0 + K], None, self._grad_views(m.bn.weight), self._grad_views(m.bn.bias), self.bn_ws, act=u.act))
                        recs.append(ops.rec_bn_slab_sums(self.bn_ws[:nb * 2 * K].view(nb, 2, K), nb, K, M, buf))
                        recs.append(self._sync_call(buf))
                        recs.append(ops.rec_bn_bwd_coeffs_sums(buf, K, c1[k0:k0 + K], c2[k0:k0 + K]))
                        k0 += K
                    recs.append(ops.rec_bn_act_bwd_apply(u.outs[0].gread(), u.outs[1].gread() if len(u.outs) > 1 else None, u.yraw, u.scale, u.shift, u.mean,
                                                         u.invstd, c1, c2, dy, act=u.act))
                if not sync and not stem_fused and PRODUCER_STATS != '0' and self.dtype == torch.bfloat16 and u.has_bn and not u.frozen and not USE_GRAPHS:
                    prod = [last.get(id(o)) for o in u.outs]
                    kinds = ('dgrad', 'fused') if PRODUCER_STATS == '1' else ('fused',)
                    ok = all(q is not None and q[0] in kinds and q[3] > 0 and o.gfinal is None and o.c % 8 == 0 and q[2] % 8 == 0 and
                             len(pending.get(q[1], [])) < 2 for q, o in zip(prod, u.outs))
                    if ok and len(u.outs) == 2 and prod[0][1] == prod[1][1]:
                        ok = len(pending.get(prod[0][1], [])) == 0
                    if not ok:
                        prod = None
                if prod is not None:
                    k0 = 0
                    for m, o, (kind, idx, off, nslabs) in zip(u.mods, u.outs, prod):
                        K = o.c
                        slabs = self._new(nslabs, 2, K, dtype=torch.float32, zero=True)     # workgroups without tiles never write theirs
                        pending.setdefault(idx, []).append(ops.StatRequest(u.yraw[..., k0:k0 + K], u.scale[k0:k0 + K], u.shift[k0:k0 + K], slabs, off, u.act))
                        recs.append(ops.rec_bn_bwd_finalize_slabs(slabs, M, u.mean[k0:k0 + K], u.invstd[k0:k0 + K], self._grad_views(m.bn.weight),
                                                                  self._grad_views(m.bn.bias), c1[k0:k0 + K], c2[k0:k0 + K]))
                        k0 += K
                    if not fused:
                        recs.append(ops.rec_bn_act_bwd_apply(u.outs[0].gread(), u.outs[1].gread() if len(u.outs) > 1 else None, u.yraw, u.scale, u.shift, u.mean,
                                                             u.invstd, c1, c2, dy, act=u.act))
                elif sync:
                    pass
                elif pair:
                    ma, mb = u.mods
                    recs.append(ops.rec_bn_act_bwd_pair(u.outs[0].gread(), u.outs[1].gread(), u.yraw, u.scale, u.shift, u.mean, u.invstd, dy,
                                                        self._grad_views(ma.bn.weight), self._grad_views(ma.bn.bias),
                                                        self._grad_views(mb.bn.weight), self._grad_views(mb.bn.bias), self.bn_ws, act=u.act))
                    if fused:
                        c1, c2 = ops.bn_bwd_coeffs(self.bn_ws, M, u.K)
                k0 = 0
                for m, o in zip(u.mods, u.outs):
                    if pair or prod is not None or sync:
                        break
                    K = m.conv.out_channels
                    dyk = None if dy is None else dy[..., k0:k0 + K]
                    if u.frozen:
                        recs.append(ops.rec_bn_act_bwd(o.gread(), u.yraw[..., k0:k0 + K], u.scale[k0:k0 + K], u.shift[k0:k0 + K], None, None,
                                                       dyk, None, None, self.bn_ws, act=u.act))
                    else:
                        recs.append(ops.rec_bn_act_bwd(o.gread(), u.yraw[..., k0:k0 + K], u.scale[k0:k0 + K], u.shift[k0:k0 + K],
                                                       u.mean[k0:k0 + K], u.invstd[k0:k0 + K], dyk,
                                                       self._grad_views(m.bn.weight), self._grad_views(m.bn.bias), self.bn_ws, act=u.act))
                        if fused or stem_fused:
                            c1, c2 = ops.bn_bwd_coeffs(self.bn_ws, M, u.K)
                    k0 += K
                x = self.prep if u.stem else u.x.t()
                stem_hw = (self.H, self.W) if u.stem else None
                want_w = any(m.conv.weight.requires_grad for m in u.mods)
                want_x = not u.stem and u.x is not self.input and up(u.x)
                acc, xv = False, u.x
                if want_x:
                    wb = u.mods[1].conv.weight if len(u.mods) > 1 else None
                    self.packs.add(u.mods[0].conv.weight, wb, u.s, u.p, ops.PACK_DGRAD, u.wpd)
                    if xv.parts is not None:
                        # writing the whole concat gradient: no part may already hold a partial contribution
                        if any(pv.ginit for pv, _ in xv.parts) and not xv.ginit:
                            raise _lib.HdyError('gradient ordering not plannable: a concat input receives gradient before the concat')
                        acc = xv.ginit
                        xv.ginit = True
                    else:
                        acc = self._contrib(xv)
                if fused:
                    if want_w or want_x:
                        ga = self._grad_views(u.mods[0].conv.weight) if want_w else None
                        gb = self._grad_views(u.mods[1].conv.weight) if want_w and len(u.mods) > 1 else None
                        mk = (lambda st, u=u, x=x, xv=xv, c1=c1, c2=c2, ga=ga, gb=gb, acc=acc, want_x=want_x: ops.rec_conv1x1_bwd_fused(
                            u.outs[0].gread(), u.outs[1].gread() if len(u.mods) > 1 else None, u.yraw, u.scale, u.shift, u.mean, u.invstd, c1, c2, x,
                            u.wpd if want_x else None, xv.g() if want_x else None, ga, gb, self.f1_ws, accumulate_dx=acc, stats=st))
                        recs.append(mk(None))
                        if want_x:
                            note_grad(xv, 'fused', len(recs) - 1, mk, ops.fused_1x1_stat_slabs(M, u.C, u.K, self.dtype))
                    continue
                ga = self._grad_views(u.mods[0].conv.weight)
                gb = self._grad_views(u.mods[1].conv.weight) if len(u.mods) > 1 else None
                if stem_fused:
                    wgrad(ops.rec_conv_wgrad_stem_fused(x, u.outs[0].gread(), u.yraw, u.scale, u.shift, u.mean, u.invstd, c1, c2, stem_hw, ga, None,
                                                        self.wg_ws))
                    continue
                if want_w:
                    wgrad(lambda ws, x=x, dy=dy, ga=ga, gb=gb, u=u, stem_hw=stem_hw: ops.rec_conv_wgrad(x, dy, ga, gb, u.k, u.k, u.s, u.p, ws, stem_hw=stem_hw),
                          reads_dy_slot=slot, pixels=M)
                if want_x:
                    mk = (lambda st, u=u, dy=dy, xv=xv, acc=acc: ops.rec_conv_dgrad(dy, u.wpd, xv.g(), u.k, u.k, u.s, u.p, accumulate=acc, stats=st))
                    recs.append(mk(None))
                    note_grad(xv, 'dgrad', len(recs) - 1, mk,
                              ops.conv_dgrad_stat_slabs(xv.n, xv.h, xv.w, xv.c, u.K, u.k, u.k, u.s, u.p, self.dtype) if self.dtype == torch.bfloat16 else 0)
        for idx, reqs in pending.items():                   # rebuild the producers with the statistics requests they serve
            recs[idx] = remake[idx](reqs)
        self.producer_stat_units = sum(len(v) for v in pending.values())
        self._mark_buckets(recs, side, nfork)
        last_fork = next((r[3] for r in reversed(recs) if r[0] == '@fork'), None)      # in LIST order (the reduce forks got their tokens later)
        if side is not None and last_fork is not None:
            recs.append(('@join', side, last_fork))             # side-stream work is in order: the last fork covers all
        return recs

    def _mark_buckets(self, recs, side, nfork=None):
        """Data-parallel overlap (reference: DDP's autograd-hook buckets, train.py:331): cut the flat gradient buffer into ranges of
        about GRAD_BUCKET_BYTES and insert, behind the launch record that completes a range, a '@call' that tells the engine so.
        Parameters are laid out in registration (= forward) order and the backward list runs in reverse, so ranges complete from the
        end of the buffer; a range's mark sits behind the LAST record writing into it whatever the order.  Gradients this list does
        not produce (frozen parameters, the mask head's, which MaskBranchFn writes before the list runs) are final from the start."""
        log, self._grad_log = self._grad_log, None
        store = self.grad_store
        done = {}
        for q, pos in log:
            done[id(q)] = max(done.get(id(q), -1), pos)
        items = sorted((store.offsets[id(q)], q) for q in store.params)
        marks, hi, acc, ready = [], store.numel, 0, -1
        for off, q in reversed(items):
            acc += q.numel() * 4
            ready = max(ready, done.get(id(q), -1))
            if acc >= GRAD_BUCKET_BYTES or off == 0:
                marks.append((ready, off, hi))
                hi, acc, ready = off, 0, -1
        # a range whose last writer comes later than that of a range cut after it can only go out with it: merge by position
        by_pos = {}
        for ready, a, b in marks:
            by_pos.setdefault(ready, []).append((a, b))
        self.grad_marks = []
        for pos in sorted(by_pos, reverse=True):              # insert from the back so earlier positions stay valid
            for a, b in by_pos[pos]:
                fn = (lambda a=a, b=b: self.bucket_hook(a, b, side.stream if side is not None else None) if self.bucket_hook else None)
                fn.hdy_mark = (a, b)            # bucket_marks() reads the marks back in execution order
                recs.insert(pos + 1, ('@call', fn))
                self.grad_marks.append((pos + 1, a, b))

    def bucket_marks(self):
        """the "flat gradient range [a, b) is final" marks of the backward list, in the order the list reaches them"""
        return [rec[1].hdy_mark for rec in (self.bwd or []) if rec[0] == '@call' and hasattr(rec[1], 'hdy_mark')]

    def _det_bias_tmp(self, u):
        if not hasattr(u, 'gbias_pad'):
            u.gbias_pad = self._new(u.Kp, dtype=torch.float32)
        return u.gbias_pad

    def fused_loss(self, head):
        """The hdy_det_loss call bound to this plan's logits / gradient buffers (built once)."""
        if self.loss_call is None:
            anc = [float(v) for buf in head.anchors for v in buf.anchor.flatten().tolist()]
            cw = head.det_loss.BCEcls.weight if hasattr(head.det_loss.BCEcls, 'weight') else None
            cw = [1.0] * head.nc if cw is None else ([float(cw)] * head.nc if cw.numel() == 1 else [float(v) for v in cw.flatten().tolist()])
            self.loss_call = ops.DetLossCall([u.logits for u in self.det_units], [u.gdet for u in self.det_units], head.na, head.nc, anc,
                                             head.det_loss.balance, cw, head.det_loss.hyp, self.loss_out, self.device)
        return self.loss_call

    def tap_features(self):
        return [self.feature(k) for k in self.tap_keys]

    def run_backward(self, gdets=None, scale=None, gtaps=None):
        """gdets: autograd's logits gradients (unfused loss), or None when the fused loss kernel already filled the plan's
        gradient buffers; then `scale` is the upstream gradient of the loss (1-element device tensor).  gtaps: gradients of the
        tapped feature maps (NCHW-shaped, or None = no outside consumer contributed in this step)."""
        if self.grad_store is not None:
            self.grad_store.before_backward()
        for i, k in enumerate(self.tap_keys):
            g = None if gtaps is None else gtaps[i]
            if g is None:
                self.outs[k].g().zero_()
            else:
                self.outs[k].g().copy_(g.permute(0, 2, 3, 1))
        if self.tap_params and not self.tap_grads_ready:
            for q in self.tap_params:
                self.grad_store.view_of(q).zero_()
        self.tap_grads_ready = False
        if gdets is None:
            if scale is not None:
                ops.scale_inplace(self.gdet_flat, scale.reshape(-1)[:1].float().contiguous())
            pre = []
        else:
            pre = []
            for u, g in zip(self.det_units, gdets):
                if g is None:
                    u.gdet.zero_()
                else:
                    pre.append(ops.rec_det_grad_pack(g, u.gdet, u.na, u.no))
        ops.run(pre)
        if self.mask_vals and not self.mask_grads_ready:
            for v in self.mask_vals:                 # no mask loss in this step: the branch contributes nothing
                v.g().zero_()
            for q in self.seg_h_params:
                self.grad_store.view_of(q).zero_()
        self.mask_grads_ready = False
        self._replay('bwd', self.bwd)
