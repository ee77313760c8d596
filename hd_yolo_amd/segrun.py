"""Executor of the semantic-segmentation branch on the HIP kernels (reference: hnet/segmentation/utils_seg.py:5-59
PanopticFeatureConnector, hnet/segmentation/panoptic_seg.py:3-43 PanopticSeg).

Per pyramid level a ladder of [conv3x3 (no bias) -> GroupNorm(32) -> ReLU (-> bilinear x2, align_corners=True)] stages brings the level to
the resolution of the finest one; the ladders' outputs are summed (the last stage of every ladder writes / accumulates straight into
the sum buffer); a 1x1 conv (+bias) gives per-pixel class logits, bilinear-resized to the mask resolution, Softmax2d + soft dice.
The reference resizes the 128-channel feature map BEFORE the 1x1 conv (panoptic_seg.py:13-19); a 1x1 conv and a bilinear resize are
both linear and commute (the resize weights sum to 1, so the bias commutes too), hence the conv runs first, on 1/s^2 of the pixels,
and the resize moves nc channels instead of 128.

Everything is NHWC; buffers are allocated per call and the launch records run immediately (as hd_yolo_amd/maskhead.py).  Backward =
reverse walk: resize^T, GroupNorm/ReLU backward, weight gradient and data gradient of each conv on the detector's conv kernels."""
import torch

from . import ops


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
        if hit is not None and hit[0] == tag and hit[2]() is weight:
            return hit[1]
        Kw, C, R, S = weight.shape
        wp = ops.pack_alloc(Kw if K is None else K, C, R, S, stride, pad, kind, dtype, weight.device)
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


class PanopticRun:
    """One forward (+ backward) of connector + class head.  feats: NHWC tensors, finest level first (the connector's order)."""

    def __init__(self, connector, class_conv, dtype, cache=None):
        self.levels = [ladder_stages(seq) for seq in connector.layers.values()]
        self.class_conv, self.dtype = class_conv, dtype
        self.cache = cache or PackCache()

    def forward(self, feats, out_size=None, train=False):
        dt = self.dtype
        self.tape = []
        total = None
        for feat, stages in zip(feats, self.levels):
            h, rec = feat, []
            for si, (conv, gn, up) in enumerate(stages):
                N, H, W, _ = h.shape
                K = conv.out_channels
                wp = self.cache.get(conv.weight, 1, 1, ops.PACK_FWD, dt)
                y = torch.empty((N, H, W, K), dtype=dt, device=h.device)
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
        low = torch.zeros((N, H, W, kp), dtype=torch.float32, device=total.device)
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

    def backward(self, dlogits, grad_of, scale=None, w_reduced=False):
        """dlogits fp32 (N, Ho, Wo, nc4 | kp) -> list of feature gradients (NHWC, finest first); parameter gradients into grad_of(p).
        scale: 1-element tensor multiplied in AFTER the resize backward (linear: the same result, on 1/64 of the elements)."""
        dt = self.dtype
        N, H, W, kp = self.low_shape
        dev = dlogits.device
        cc = self.class_conv
        nc = cc.out_channels
        if self.out_size == (H, W):
            dlow = dlogits
        elif w_reduced:                                        # (N, Ho, W, 4): the loss kernel already ran the W pass of the resize backward
            dlow = torch.zeros((N, H, W, kp), dtype=torch.float32, device=dev)
            ops.bilinear_bwd_h(dlogits, H, dlow[..., :dlogits.shape[3]])
        else:
            dlow = torch.zeros((N, H, W, kp), dtype=torch.float32, device=dev)
            ops.bilinear_bwd(dlogits, (H, W), out=dlow[..., :dlogits.shape[3]])
        if scale is not None:
            dlow = dlow * scale.reshape(-1)[:1].to(dlow.dtype)
        g = dlow.to(dt)
        ws_bn = torch.empty(ops.bn_bwd_ws_floats(N * H * W, max(kp, 8)), dtype=torch.float32, device=dev)
        tmp = torch.empty(kp, dtype=torch.float32, device=dev)
        ops.run([ops.rec_colsum(g, tmp, ws_bn)])
        grad_of(cc.bias).copy_(tmp[:nc])
        C = self.total.shape[3]
        gw = torch.empty((kp, C, 1, 1), dtype=torch.float32, device=dev)
        ws = torch.empty(ops.wgrad_ws_bytes(N, H, W, C, kp, 1, 1, 1, 0, dt) // 4 + 16, dtype=torch.float32, device=dev)
        ops.run([ops.rec_conv_wgrad(self.total, g, gw, None, 1, 1, 1, 0, ws)])
        grad_of(cc.weight).copy_(gw[:nc])
        wl_d = self.cache.get(cc.weight, 1, 0, ops.PACK_DGRAD, dt, K=kp)
        dtotal = torch.empty_like(self.total)
        ops.run([ops.rec_conv_dgrad(g, wl_d, dtotal, 1, 1, 1, 0)])
        dfeats = []
        for rec in self.tape:
            dh = dtotal
            for conv, gn, up, xin, y, saved, last in reversed(rec):
                Nn, Hh, Ww, K = y.shape
                dz = ops.bilinear_bwd(dh, (Hh, Ww)) if up else dh
                dy = ops.groupnorm_relu_bwd(dz, y, gn.weight.detach().float(), saved, gn.num_groups, grad_of(gn.weight), grad_of(gn.bias))
                Cin = xin.shape[3]
                wsz = torch.empty(ops.wgrad_ws_bytes(Nn, Hh, Ww, Cin, K, 3, 3, 1, 1, dt) // 4 + 16, dtype=torch.float32, device=dev)
                ops.run([ops.rec_conv_wgrad(xin, dy, grad_of(conv.weight), None, 3, 3, 1, 1, wsz)])
                wd = self.cache.get(conv.weight, 1, 1, ops.PACK_DGRAD, dt)
                dh = torch.empty((Nn, Hh, Ww, Cin), dtype=dt, device=dev)
                ops.run([ops.rec_conv_dgrad(dy, wd, dh, 3, 3, 1, 1)])
            dfeats.append(dh)
        self.tape = None
        return dfeats
