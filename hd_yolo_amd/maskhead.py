"""The Mask-RCNN head of the mask branch on the HIP kernels (reference: metayolo/models/yolo_head.py:123-128
`build_seg_layers`, torchvision MaskRCNNHeads = 4 x [conv3x3 + bias + ReLU], MaskRCNNPredictor = ConvTranspose2d(2, 2) + ReLU +
conv1x1 + bias).

The number of rois changes every step, so this is not part of the static plan: buffers are allocated per call and launch records
run immediately.  Everything is NHWC: roi features (R, 14, 14, C) -> four conv3x3 -> deconv (as the stride-2 dgrad of the
equivalent 2x2/s2 convolution) -> (R, 28, 28, 256) -> conv1x1 -> fp32 logits (R, 28, 28, nc_masks).
Backward = ReLU mask, bias column sums, wgrad and dgrad of each layer, gradients written into the engine's flat store."""
import torch

from . import ops


_CACHE = None


def _pack(weight, stride, pad, kind, dtype):
    """Packed kernel operand of a head weight, re-packed only when the parameter changed (version counter / storage): evaluation
    loops and the several uses of one weight inside a training step (forward, data gradient) share one packing."""
    global _CACHE
    if _CACHE is None:
        from .segrun import PackCache
        _CACHE = PackCache()
    return _CACHE.get(weight, stride, pad, kind, dtype)


class MaskHeadRun:
    """One forward (and optionally backward) of seg_h on a batch of roi features."""

    def __init__(self, seg_h, dtype):
        heads, preds = seg_h.maskrcnn_heads, seg_h.maskrcnn_preds
        self.convs = [m for m in heads.children() if isinstance(m, torch.nn.Conv2d)]
        self.deconv, self.logits = preds.conv5_mask, preds.mask_fcn_logits
        self.dtype = dtype
        for c in self.convs:
            assert c.kernel_size == (3, 3) and c.stride == (1, 1) and c.padding == (1, 1) and c.dilation == (1, 1)
        assert self.deconv.kernel_size == (2, 2) and self.deconv.stride == (2, 2) and self.deconv.padding == (0, 0)

    # ---- forward: returns fp32 logits (R, 28, 28, nc_masks); keeps what the backward needs when `train`
    def forward(self, x, train=False):
        dt, dev = self.dtype, x.device
        R, P = x.shape[0], x.shape[1]
        self.acts = [x]
        h = x
        for c in self.convs:
            wp = _pack(c.weight, 1, 1, ops.PACK_FWD, dt)
            y = torch.empty((R, P, P, c.out_channels), dtype=dt, device=dev)
            ops.run([ops.rec_conv_fwd(h, wp, y, c.out_channels, 3, 3, 1, 1, shift=c.bias.detach().float(), act=ops.ACT_RELU)])
            self.acts.append(y)
            h = y
        # ConvTranspose2d(Cin, Cout, 2, 2): weight [Cin][Cout][2][2] == the weight of a Conv2d(Cout -> Cin, 2x2, stride 2); its
        # transpose-conv is that convolution's data gradient
        d = self.deconv
        wd = _pack(d.weight, 2, 0, ops.PACK_DGRAD, dt)
        up_raw = torch.empty((R, 2 * P, 2 * P, d.out_channels), dtype=dt, device=dev)
        ops.run([ops.rec_conv_dgrad(h, wd, up_raw, 2, 2, 2, 0)])
        up = torch.empty_like(up_raw)
        one = torch.ones(d.out_channels, dtype=torch.float32, device=dev)
        ops.run([ops.rec_bn_act_fwd(up_raw, one, d.bias.detach().float(), up, act=ops.ACT_RELU)])
        self.acts.append(up)
        lg = self.logits
        wl = _pack(lg.weight, 1, 0, ops.PACK_FWD, dt)
        kp = (lg.out_channels + 7) // 8 * 8                      # channel pitch of the fp32 logits
        out = torch.zeros((R, 2 * P, 2 * P, kp), dtype=torch.float32, device=dev)
        ops.run([ops.rec_conv_fwd(up, wl, out[..., :lg.out_channels], lg.out_channels, 1, 1, 1, 0, shift=lg.bias.detach().float())])
        if not train:
            self.acts = None
        return out[..., :lg.out_channels]

    # ---- backward: dlogits fp32 (R, 28, 28, nc_masks) -> gradient of the roi features; parameter gradients into grad_of(p)
    def backward(self, dlogits, grad_of):
        dt, dev = self.dtype, dlogits.device
        x0, up = self.acts[0], self.acts[-1]
        R, P = x0.shape[0], x0.shape[1]
        lg, d = self.logits, self.deconv
        K = lg.out_channels
        kp = (K + 7) // 8 * 8
        g = torch.zeros((R, 2 * P, 2 * P, kp), dtype=dt, device=dev)
        g[..., :K] = dlogits.to(dt)
        M2 = R * 4 * P * P
        ws_bn = torch.empty(ops.bn_bwd_ws_floats(M2, 256), dtype=torch.float32, device=dev)

        def wgrad(xin, dy, weight, R_, S_, stride, pad, Kpad=None):
            N, H, W, C = xin.shape
            Kd = dy.shape[3]
            ws = torch.empty(ops.wgrad_ws_bytes(N, H, W, C, Kd, R_, S_, stride, pad, dt) // 4 + 16, dtype=torch.float32, device=dev)
            gw = grad_of(weight)
            if Kpad is None:
                ops.run([ops.rec_conv_wgrad(xin, dy, gw, None, R_, S_, stride, pad, ws)])
            else:                                   # dy carries zero padding channels: gradient rows beyond K are dropped
                tmp = torch.empty((Kpad,) + tuple(weight.shape[1:]), dtype=torch.float32, device=dev)
                ops.run([ops.rec_conv_wgrad(xin, dy, tmp, None, R_, S_, stride, pad, ws)])
                gw.copy_(tmp[:weight.shape[0]])

        def bias_grad(dy, bias, n):
            tmp = torch.empty(dy.shape[3], dtype=torch.float32, device=dev)
            ops.run([ops.rec_colsum(dy, tmp, ws_bn)])
            grad_of(bias).copy_(tmp[:n])

        # logits conv 1x1
        bias_grad(g, lg.bias, K)
        wgrad(up, g, lg.weight, 1, 1, 1, 0, Kpad=kp)
        wl_d = ops.pack_alloc(kp, up.shape[3], 1, 1, 1, 0, ops.PACK_DGRAD, dt, dev)
        ops.run([ops.rec_pack(lg.weight.detach().float(), None, 1, 0, ops.PACK_DGRAD, wl_d, K=kp)])
        dup = torch.empty_like(up)
        ops.run([ops.rec_conv_dgrad(g, wl_d, dup, 1, 1, 1, 0)])
        # deconv + ReLU
        du = ops.relu_bwd(dup, up)
        bias_grad(du, d.bias, d.out_channels)
        h = self.acts[-2]
        # weight gradient of the equivalent conv (input = du, output gradient = h) has the ConvTranspose2d layout [Cin][Cout][2][2]
        wgrad(du, h, d.weight, 2, 2, 2, 0)
        wc = _pack(d.weight, 2, 0, ops.PACK_FWD, dt)
        dh = torch.empty_like(h)
        ops.run([ops.rec_conv_fwd(du, wc, dh, d.in_channels, 2, 2, 2, 0)])
        # the four conv3x3 + ReLU
        for i in range(len(self.convs) - 1, -1, -1):
            c = self.convs[i]
            y, xin = self.acts[i + 1], self.acts[i]
            du = ops.relu_bwd(dh, y)
            bias_grad(du, c.bias, c.out_channels)
            wgrad(xin, du, c.weight, 3, 3, 1, 1)
            wd = _pack(c.weight, 1, 1, ops.PACK_DGRAD, dt)
            dh = torch.empty_like(xin)
            ops.run([ops.rec_conv_dgrad(du, wd, dh, 3, 3, 1, 1)])
        self.acts = None
        return dh
