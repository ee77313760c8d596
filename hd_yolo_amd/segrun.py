"""Executor of the semantic-segmentation branch on the HIP kernels (reference: hnet/segmentation/utils_seg.py:5-59
PanopticFeatureConnector, hnet/segmentation/panoptic_seg.py:3-43 PanopticSeg).

Per pyramid level a ladder of [conv3x3 (no bias) -> GroupNorm(32) -> ReLU (-> bilinear x2, align_corners=True)] stages brings the level to
the resolution of the finest one; the ladders' outputs are summed (the last stage of every ladder writes / accumulates straight into
the sum buffer); a 1x1 conv (+bias) gives per-pixel class logits, bilinear-resized to the mask resolution, Softmax2d + soft dice.
The reference resizes the 128-channel feature map BEFORE the 1x1 conv (panoptic_seg.py:13-19); a 1x1 conv and a bilinear resize are
both linear and commute (the resize weights sum to 1, so the bias commutes too), hence the conv runs first, on 1/s^2 of the pixels,
and the resize moves nc channels instead of 128.

Everything is NHWC.  Backward = reverse walk: resize^T, GroupNorm/ReLU backward, weight gradient and data gradient of each conv on the
detector's conv kernels.

Round 5: the training forward and backward are TAPED (ops.Tape).  The branch's shapes are fixed (whole-tile rois, pyramid levels of a static
plan), so the first step runs the eager code below once while a tape records its launches and keeps its buffers; every later step with the
same inputs (same feature-map addresses, parameters, gradient views) replays the tape as one compiled launch list — one C call instead of
~60 ctypes calls and ~40 allocations per step (round 4: 9.7 ms of host time per 14.7 ms hnet step, and a step that stretched to 18 ms whenever
the host was slowed, e.g. under the kernel tracer).  Inference and anything whose addresses change from call to call stay eager."""
import os

import torch

from . import ops

TAPE = os.environ.get('HDY_SEG_TAPE', '1') != '0'        # HDY_SEG_TAPE=0: the eager path everywhere (A/B, debugging)


class PackCache:
    """Packed kernel operands of module weights, re-packed only when the parameter's version counter or storage moved (every
    optimizer step bumps the version; eval loops reuse the packing)."""

    def __init__(self):
        self.slots = {}

    def _drop(self, wid):
        for key in [k for k in self.slots if k[0] == wid]:
            del self.slots[key]

    def get(self, weight, stride, pad, kind, dtype, K=None):
        key = (id(weight), kind, dtype, K)
        tag = (weight._version, weight.data_ptr())
        hit = self.slots.get(key)
        # a slot belongs to ONE live parameter object: the weak reference both frees the packed buffer when the parameter dies and keeps a
        # new parameter that happens to reuse the id (and storage address, and version 0) from inheriting another model's packing
        live = hit is not None and hit[2]() is weight
        if live and hit[0] == tag and ops.Tape.current is None:
            return hit[1]
        Kw, C, R, S = weight.shape
        # under a tape the packing launch must be ON the tape (a replay runs after an optimizer step: the weights have changed) and must
        # write the buffer the recorded convolutions read: re-pack into the slot's buffer, whatever the version says
        wp = hit[1] if (live and ops.Tape.current is not None) else ops.pack_alloc(Kw if K is None else K, C, R, S, stride, pad, kind, dtype, weight.device)
        ops.run([ops.rec_pack(weight.detach().float().contiguous(), None, stride, pad, kind, wp, K=K)])
        import weakref
        wid = id(weight)
        self.slots[key] = (tag, wp, weakref.ref(weight, lambda _r, wid=wid: self._drop(wid)))
        return wp


def ladder_stages(seq):
    """Sequential of a PanopticFeatureConnector level -> [(conv, groupnorm, upsample?)]"""
    mods, out, i = list(seq.children()), [], 0
    while i < len(mods):
        conv, gn, relu = mods[i:i + 3]
        assert isinstance(conv, torch.nn.Conv2d) and isinstance(gn, torch.nn.GroupNorm) and isinstance(relu, torch.nn.ReLU)
        assert conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.bias is None
        i += 3
        up = False
        if i < len(mods) and isinstance(mods[i], torch.nn.Upsample):
            u = mods[i]
            assert u.mode == 'bilinear' and u.align_corners and float(u.scale_factor) == 2.0
            up, i = True, i + 1
        out.append((conv, gn, up))
    return out


MAX_TAPE_MISSES = 3


class PanopticRun:
    """One forward (+ backward) of connector + class head.  feats: NHWC tensors, finest level first (the connector's order)."""

    def __init__(self, connector, class_conv, dtype, cache=None):
        self.levels = [ladder_stages(seq) for seq in connector.layers.values()]
        self.class_conv, self.dtype = class_conv, dtype
        self.cache = cache or PackCache()

    def _params(self):
        ps = [q for stages in self.levels for conv, gn, _ in stages for q in (conv.weight, gn.weight, gn.bias)]
        return ps + [self.class_conv.weight, self.class_conv.bias]

    def forward(self, feats, out_size=None, train=False):
        """Training calls are taped: the first call with a given set of input addresses runs forward_eager under a tape, later ones replay it.
        The executor is shared by every call of its header (activation buffers, tapes): ONE forward per backward.  `gen` counts the forwards;
        the autograd node remembers the one it belongs to and `check_generation` refuses a backward that another forward has overtaken."""
        self.gen = self.__dict__.get('gen', 0) + 1
        if not (TAPE and train) or self.__dict__.get('_no_tape') or any(q.dtype != torch.float32 for q in self._params()):
            self._fwd = self._bwd = None
            return self.forward_eager(feats, out_size, train)
        key = (tuple((f.data_ptr(), tuple(f.shape), tuple(f.stride())) for f in feats), None if out_size is None else tuple(out_size),
               tuple(q.data_ptr() for q in self._params()))
        ent = self.__dict__.get('_fwd')
        if ent is None or ent['key'] != key:
            # a caller whose inputs / gradient sinks are new tensors every step (a torch backbone, `.grad` re-created by zero_grad(set_to_none=True))
            # can never hit the key: after MAX_TAPE_MISSES re-recordings in a row the run goes back to plain eager launches for good
            self._misses = self.__dict__.get('_misses', 0) + (1 if ent is not None else 0)
            if self._misses >= MAX_TAPE_MISSES:
                self._no_tape, self._fwd, self._bwd = True, None, None
                return self.forward_eager(feats, out_size, True)
            tape = ops.Tape()
            with tape:
                logits = self.forward_eager(feats, out_size, True)
            ent = self._fwd = {'key': key, 'tape': tape, 'logits': logits, 'state': (self.tape, self.total, self.low_shape, self.out_size)}
            self._bwd = None
        else:
            self._misses = 0
            ent['tape'].replay()
            self.tape, self.total, self.low_shape, self.out_size = ent['state']
        return ent['logits']

    def check_generation(self, gen):
        if gen != self.__dict__.get('gen'):
            raise RuntimeError('hd_yolo_amd PanopticSeg: backward() of a segmentation forward that a later forward of the same header has overtaken '
                               f'(forward #{gen}, the executor is at #{self.__dict__.get("gen")}).  The header keeps ONE set of activation buffers per '
                               'arithmetic type: run backward before the next forward (training, eval or _logits) of this header.')

    def backward(self, dlogits, grad_of, scale=None, w_reduced=False):
        ent = self.__dict__.get('_fwd')
        if ent is None or scale is None:                       # untaped forward: eager
            return self.backward_eager(dlogits, grad_of, scale, w_reduced)
        if self.__dict__.get('_scale') is None:
            self._scale = torch.ones(1, dtype=torch.float32, device=dlogits.device)
        self._scale.copy_(scale.reshape(-1)[:1])                # the upstream factor at a fixed address (the one host-side op of a replayed step)
        key = (ent['key'], dlogits.data_ptr(), tuple(dlogits.shape), bool(w_reduced), tuple(grad_of(q).data_ptr() for q in self._params()))
        bw = self.__dict__.get('_bwd')
        if bw is None or bw['key'] != key:
            self._misses = self.__dict__.get('_misses', 0) + (1 if bw is not None else 0)
            tape = ops.Tape()
            with tape:
                dfeats = self.backward_eager(dlogits, grad_of, self._scale, w_reduced, keep=True)
            self._bwd = {'key': key, 'tape': tape, 'dfeats': dfeats}
        else:
            bw['tape'].replay()
        return self._bwd['dfeats']

    def forward_eager(self, feats, out_size=None, train=False):
        dt = self.dtype
        self.tape = []
        total = None
        for feat, stages in zip(feats, self.levels):
            h, rec = feat, []
            for si, (conv, gn, up) in enumerate(stages):
                N, H, W, _ = h.shape
                K = conv.out_channels
                wp = self.cache.get(conv.weight, 1, 1, ops.PACK_FWD, dt)
                y = ops._new((N, H, W, K), dt, h.device)
                ops.run([ops.rec_conv_fwd(h, wp, y, K, 3, 3, 1, 1)])
                z, saved = ops.groupnorm_relu_fwd(y, gn.weight.detach().float(), gn.bias.detach().float(), gn.num_groups, gn.eps)
                last = si == len(stages) - 1
                if not up:
                    nxt = z
                    if last:                                   # the finest level: its GroupNorm output starts the sum
                        assert total is None, 'only the first (finest) level may end without an upsampling stage'
                        total = z
                elif last:
                    assert total is not None and tuple(total.shape[1:3]) == (2 * H, 2 * W)
                    nxt = ops.bilinear_fwd(z, (2 * H, 2 * W), out=total, accumulate=True)
                else:
                    nxt = ops.bilinear_fwd(z, (2 * H, 2 * W))
                rec.append((conv, gn, up, h, y, saved, last))
                h = nxt
            self.tape.append(rec)
        cc = self.class_conv
        nc = cc.out_channels
        kp = (nc + 7) // 8 * 8
        N, H, W, C = total.shape
        wl = self.cache.get(cc.weight, 1, 0, ops.PACK_FWD, dt)
        low = ops._new((N, H, W, kp), torch.float32, total.device, zero=True)
        ops.run([ops.rec_conv_fwd(total, wl, low[..., :nc], nc, 1, 1, 1, 0, shift=cc.bias.detach().float())])
        out_size = (H, W) if out_size is None else tuple(out_size)
        # the resized logits are the largest tensor of the branch (mask resolution): they carry the classes padded to one 16-byte vector
        # only (4 floats for <= 4 classes), not to the conv's 8 — at 3 classes half the bytes of the resize, the dice passes and their backward
        nc4 = (nc + 3) // 4 * 4
        logits = low if out_size == (H, W) else ops.bilinear_fwd(low[..., :nc4], out_size)
        self.total, self.low_shape, self.out_size = total, (N, H, W, kp), out_size
        if not train:
            self.tape = None
        return logits                                          # fp32 NHWC (N, Ho, Wo, nc4 or kp), channels [nc, ..) are zero

    def backward_eager(self, dlogits, grad_of, scale=None, w_reduced=False, keep=False):
        """dlogits fp32 (N, Ho, Wo, nc4 | kp) -> list of feature gradients (NHWC, finest first); parameter gradients into grad_of(p).
        scale: 1-element tensor multiplied in AFTER the resize backward (linear: the same result, on 1/64 of the elements).
        Every step between the launches is itself a launch record (no tensor expressions): the sequence can run under a tape."""
        dt = self.dtype
        N, H, W, kp = self.low_shape
        dev = dlogits.device
        cc = self.class_conv
        nc = cc.out_channels
        if self.out_size == (H, W):
            dlow = ops._new((N, H, W, kp), torch.float32, dev)          # a copy: the scaling below is in place
            ops.run([ops.rec_copy_f32(dlogits.reshape(-1), dlow.reshape(-1))])
        elif w_reduced:                                        # (N, Ho, W, 4): the loss kernel already ran the W pass of the resize backward
            dlow = ops._new((N, H, W, kp), torch.float32, dev, zero=True)
            ops.bilinear_bwd_h(dlogits, H, dlow[..., :dlogits.shape[3]])
        else:
            dlow = ops._new((N, H, W, kp), torch.float32, dev, zero=True)
            ops.bilinear_bwd(dlogits, (H, W), out=dlow[..., :dlogits.shape[3]])
        if scale is not None:
            ops.scale_inplace(dlow, scale.reshape(-1)[:1].float())
        g = ops._new((N, H, W, kp), dt, dev)
        ops.cast_store(dlow, g)
        ws_bn = ops._new((ops.bn_bwd_ws_floats(N * H * W, max(kp, 8)),), torch.float32, dev)
        tmp = ops._new((kp,), torch.float32, dev)
        ops.run([ops.rec_colsum(g, tmp, ws_bn)])
        ops.run([ops.rec_copy_f32(tmp[:nc], grad_of(cc.bias))])
        C = self.total.shape[3]
        gw = ops._new((kp, C, 1, 1), torch.float32, dev)
        ws = ops._new((ops.wgrad_ws_bytes(N, H, W, C, kp, 1, 1, 1, 0, dt) // 4 + 16,), torch.float32, dev)
        ops.run([ops.rec_conv_wgrad(self.total, g, gw, None, 1, 1, 1, 0, ws)])
        ops.run([ops.rec_copy_f32(gw[:nc].reshape(-1), grad_of(cc.weight).reshape(-1))])
        wl_d = self.cache.get(cc.weight, 1, 0, ops.PACK_DGRAD, dt, K=kp)
        dtotal = ops._new(tuple(self.total.shape), self.total.dtype, dev)
        ops.run([ops.rec_conv_dgrad(g, wl_d, dtotal, 1, 1, 1, 0)])
        dfeats = []
        for rec in self.tape:
            dh = dtotal
            for conv, gn, up, xin, y, saved, last in reversed(rec):
                Nn, Hh, Ww, K = y.shape
                dz = ops.bilinear_bwd(dh, (Hh, Ww)) if up else dh
                dy = ops.groupnorm_relu_bwd(dz, y, gn.weight.detach().float(), saved, gn.num_groups, grad_of(gn.weight), grad_of(gn.bias))
                Cin = xin.shape[3]
                wsz = ops._new((ops.wgrad_ws_bytes(Nn, Hh, Ww, Cin, K, 3, 3, 1, 1, dt) // 4 + 16,), torch.float32, dev)
                ops.run([ops.rec_conv_wgrad(xin, dy, grad_of(conv.weight), None, 3, 3, 1, 1, wsz)])
                wd = self.cache.get(conv.weight, 1, 1, ops.PACK_DGRAD, dt)
                dh = ops._new((Nn, Hh, Ww, Cin), dt, dev)
                ops.run([ops.rec_conv_dgrad(dy, wd, dh, 3, 3, 1, 1)])
            dfeats.append(dh)
        if not keep:
            self.tape = None
        return dfeats
