"""Per-kernel parity tests on a real MI355X, through the C ABI (hd_yolo_amd/ops.py -> libhdyolo_hip.so).

Floating-point kernels are compared with a plain PyTorch fp32 CPU evaluation of the same op; decode with the
golden vectors made by the reference; NMS bit-exactly with the CPU oracle (oracle/nms_ref).
Tolerances: fp32 mode 1e-4 relative to the tensor's max magnitude (north star: 1e-4 relative);
bf16 mode 1.5e-2 (operands are pre-rounded to bf16 on both sides, so only accumulation order and the final
bf16 rounding of outputs differ: 2^-8 = 3.9e-3 per rounding).
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from hd_yolo_amd import _lib, ops, synth  # noqa: E402

DEV = 'cuda:0'
TOL = {torch.float32: 1e-4, torch.bfloat16: 1.5e-2}
DTYPES = [torch.float32, torch.bfloat16]


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


def q(t, dtype):
    """round through the arithmetic type (what the kernel sees)"""
    return t.to(dtype).float()


def to_dev_nhwc(t_nchw, dtype, ld=None, off=0):
    """NCHW cpu fp32 -> NHWC device tensor of `dtype`, optionally a channel slice [off:off+C] of a pitch-ld buffer."""
    n, c, h, w = t_nchw.shape
    ld = ld or c
    buf = torch.full((n, h, w, ld), 7.0, dtype=dtype, device=DEV)     # poison outside the slice
    view = buf[..., off:off + c]
    view.copy_(t_nchw.permute(0, 2, 3, 1).to(dtype))
    return view


def from_dev_nhwc(v):
    return v.float().cpu().permute(0, 3, 1, 2).contiguous()


def assert_close(got, ref, tol, what='', elementwise=None):
    """Two criteria.  (1) max |error| / max |ref| < tol.  (2) for the fp32 path (tol <= 3e-4; north_star: "within 1e-4 relative") EVERY element:
    |error| <= tol * |ref| + 0.2 * tol * rms(ref) — the absolute term is what a sum of ~1000 fp32 products in a different order costs an element
    that cancels to near zero (measured: <= 0.1 * tol * rms on every tensor of the model goldens, scripts/probes/elementwise_parity.py)."""
    scale = ref.abs().max().item() + 1e-12
    d = (got - ref).abs()
    err = d.max().item() / scale
    assert err < tol, f'{what}: max rel-to-max error {err:.3e} >= {tol}'
    if tol <= 3e-4 if elementwise is None else elementwise:
        rms = ref.float().pow(2).mean().sqrt().item()
        over = (d - tol * ref.abs() - 0.2 * tol * rms).max().item()
        assert over <= 0, f'{what}: an element is off by {over + 0.2 * tol * rms:.3e} beyond {tol} * |ref| (allowed absolute part {0.2 * tol * rms:.3e})'


CONV_CASES = [
    # N, H, W, C, K, R, stride, pad
    (2, 20, 20, 64, 32, 1, 1, 0),
    (1, 16, 16, 128, 256, 1, 1, 0),
    (2, 20, 20, 32, 32, 3, 1, 1),
    (1, 17, 13, 64, 64, 3, 1, 1),
    (1, 10, 10, 48, 24, 3, 1, 1),
    (2, 20, 20, 32, 64, 3, 2, 1),
    (1, 18, 14, 64, 128, 3, 2, 1),
    (1, 17, 15, 32, 64, 3, 2, 1),        # odd sides: the four parity classes differ in size -> four launches instead of the class walk
    (3, 64, 96, 64, 128, 3, 2, 1),       # class walk over many tiles: workgroup ranges start mid-group
    (2, 48, 48, 128, 256, 3, 2, 1),      # two column tiles per class
    (3, 40, 40, 256, 512, 1, 1, 0),
    (2, 32, 64, 32, 64, 3, 2, 1),        # the patch-resident stride-2 data gradient (dx 32 <- dy 64 channels, dy grid 16 x 32 = 2 x 2 tiles)
    (2, 32, 64, 64, 128, 3, 2, 1),       # ... its 64 <- 128 form (two dy planes, eight waves): dy grid 16 x 32 = 2 x 2 tiles
    (2, 24, 24, 16, 32, 3, 2, 1),        # stride-2 dgrad over 32 gradient channels: class walk with k-blocks that straddle taps
    (1, 20, 28, 32, 48, 3, 2, 1),        # ... over 48
    (2, 20, 20, 128, 128, 3, 1, 1),      # wide 3x3 / stride 1: one column tile
    (3, 40, 40, 128, 256, 3, 1, 1),      # ... two column tiles (interleaved tile order), tiles that straddle images
    (1, 24, 56, 192, 128, 3, 1, 1),      # ... three channel blocks
    (2, 16, 32, 64, 64, 3, 1, 1),        # qualifies for the filter-resident 3x3 kernel (bf16): C=64, H%8==0, W%16==0
    (3, 32, 48, 64, 32, 3, 1, 1),
    (2, 24, 32, 64, 64, 3, 1, 1),        # H % 16 != 0
    (2, 16, 32, 32, 32, 3, 1, 1),        # the same kernel with 32 input channels (64-byte patch rows, one k-step per tap)
    (3, 40, 48, 32, 64, 3, 1, 1),
    (5, 128, 128, 32, 32, 3, 1, 1),      # band split, 32 channels
    # filter-resident kernel, more 8x16 tiles than its 512 workgroups: 640 tiles = 1 whole + a 2-row band each; 704 -> 4-row bands; 832 -> whole
    (5, 128, 128, 64, 64, 3, 1, 1),
    (2, 176, 256, 64, 48, 3, 1, 1),
    (2, 208, 256, 64, 64, 3, 1, 1),
    # the filter-resident 3x3 kernel for 128 input channels (round 6, conv3x3_c128.hip): 8 x 8 pixel items, the two channel halves on neighbouring workgroups
    (2, 40, 40, 128, 128, 3, 1, 1),      # yolov5s' bottleneck shape: 50 items per half on 100 workgroups
    (3, 24, 40, 128, 128, 3, 1, 1),      # 45 tiles: streams with one and with two items
    (1, 8, 8, 128, 128, 3, 1, 1),        # a single tile: every patch pixel on the border
    (2, 16, 24, 128, 64, 3, 1, 1),       # one channel half
    (2, 16, 16, 128, 104, 3, 1, 1),      # ragged second half (K = 64 + 40)
    (9, 64, 64, 128, 128, 3, 1, 1),      # 576 tiles on 256 streams: three items on some, two on others
    # yolov5m's widths (round 6; yolov5m.yaml:5-6, yolov5.py:127 make_divisible(c * 0.75, 8)): 48 / 96 / 192 / 384 — rows that are not multiples of 128
    # bytes, k-blocks that straddle taps (C % 64 != 0), column tiles that are partly empty (K = 96 on a 128-wide tile, K = 192 on two)
    (2, 32, 32, 48, 48, 3, 1, 1),        # the first C3's bottleneck 3x3
    (2, 32, 32, 48, 48, 1, 1, 0),
    (2, 32, 32, 48, 96, 3, 2, 1),        # backbone.1
    (2, 24, 24, 96, 96, 3, 1, 1),
    (2, 24, 24, 96, 96, 1, 1, 0),
    (2, 24, 24, 96, 192, 3, 2, 1),       # backbone.3
    (2, 20, 20, 192, 192, 3, 1, 1),
    (2, 20, 20, 192, 192, 1, 1, 0),
    (2, 24, 24, 192, 192, 3, 2, 1),      # neck downsample: the deep-pipelined weight gradient from K = 192 (round 6)
    (2, 20, 20, 192, 384, 3, 2, 1),      # backbone.5
    (1, 20, 20, 384, 384, 3, 1, 1),
    (2, 20, 20, 384, 192, 1, 1, 0),      # neck lateral
    (1, 20, 20, 384, 768, 3, 2, 1),      # backbone.7
    (1, 20, 20, 768, 384, 1, 1, 0),
]


# kernel family the dispatch log must name for a CONV_CASES row in bf16 (forward with statistics, data gradient, weight gradient); None = not
# asserted.  A regression that silently sends one of these shapes back to the generic kernel passes every numeric check.
def expected_dispatch(case):
    N, H, W, C, K, R, stride, pad = case
    fwd = dgrad = wgrad = None
    if R == 3 and stride == 1 and pad == 1 and C in (32, 64) and K <= 64 and H % 8 == 0 and W % 16 == 0:
        fwd = 'conv3x3_c%d' % C
        if K in (32, 64) and C <= 64:
            dgrad = 'conv3x3_c%d' % K
    if R == 3 and stride == 1 and pad == 1 and C == 128 and K <= 128 and H % 8 == 0 and W % 8 == 0:
        fwd = 'conv3x3_c128'
    if R == 3 and stride == 1 and pad == 1 and K == 128 and C <= 128 and C % 8 == 0 and H % 8 == 0 and W % 8 == 0:
        dgrad = 'conv3x3_c128'             # the data gradient is the same convolution with the roles of C and K swapped
    if R == 3 and stride == 2 and C == 32 and K == 64 and H % 16 == 0 and W % 64 == 0:
        fwd, dgrad = 'conv3x3s2_c32', 'dgrad3x3s2_k64c32'
    if R == 3 and stride == 2 and C == 64 and K == 128 and H % 8 == 0 and W % 32 == 0:
        fwd, dgrad = 'conv3x3s2_c64', 'dgrad3x3s2_k128c64'
    if R == 3 and stride == 1 and C % 32 == 0 and K % 32 == 0 and C <= 256 and K <= 256 and (C % 64 == 0 or C == 32) and (K % 64 == 0 or K == 32):
        wgrad = 'wgrad3x3'                 # (several 32-wide blocks per split — 96 x 96 — measured slower than the generic kernel: conv_wgrad3x3.hip w3_plan)
    if R == 3 and stride == 2 and C % 64 == 0 and K % 64 == 0 and K >= 192 and N * (H // 2) * (W // 2) >= 8192:
        wgrad = 'wgrad_deep'
    return fwd, dgrad, wgrad


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_fwd_dgrad_wgrad(case, dtype):
    logs = conv_case(case, dtype)
    if dtype == torch.bfloat16:
        for got, want, what in zip(logs, expected_dispatch(case), ('forward', 'data gradient', 'weight gradient')):
            if want is not None:
                assert want in got, f'{what} of {case} ran {got}, expected {want}'


def conv_case(case, dtype):
    """forward (+ statistics), epilogue, data gradient, weight gradient of one convolution against torch fp32 on the CPU; returns the kernel
    families the launchers picked for (forward with statistics, data gradient, weight gradient)"""
    N, H, W, C, K, R, stride, pad = case
    x = q(rnd((N, C, H, W), 1), dtype)
    w = rnd((K, C, R, R), 2, (3.0 / (C * R * R)) ** 0.5)
    wq = q(w, dtype)
    xd = to_dev_nhwc(x, dtype, ld=C + 16, off=8)
    Ho, Wo = ops.out_dim(H, R, stride, pad), ops.out_dim(W, R, stride, pad)
    wdev = w.to(DEV)
    wp = ops.pack_alloc(K, C, R, R, stride, pad, ops.PACK_FWD, dtype, DEV)
    ybuf = torch.full((N, Ho, Wo, K + 8), 5.0, dtype=dtype, device=DEV)
    y = ybuf[..., 8:]
    mt = ops.stat_slabs(N, H, W, C, K, R, R, stride, pad, dtype)
    stats = torch.full((mt, 2, K), float('nan'), dtype=torch.float32, device=DEV)       # every slab must be written
    ops.run([ops.rec_pack(wdev, None, stride, pad, ops.PACK_FWD, wp)])
    _lib.dispatch_log(reset=True)
    ops.run([ops.rec_conv_fwd(xd, wp, y, K, R, R, stride, pad, stats=stats)])
    log_fwd = _lib.dispatch_log(reset=True)
    ref = F.conv2d(x, wq, None, stride, pad)
    got = from_dev_nhwc(y)
    assert_close(got, ref, TOL[dtype], 'conv fwd')
    assert (ybuf[..., :8].float() == 5.0).all(), 'wrote outside the channel slice'
    s = stats.sum(0).cpu()
    assert_close(s[0], ref.sum((0, 2, 3)), 1e-3, 'stats sum')
    assert_close(s[1], (ref * ref).sum((0, 2, 3)), 1e-3, 'stats sumsq')

    # epilogue: scale/shift + SiLU + accumulate
    sc, sh = rnd((K,), 3).abs() + 0.5, rnd((K,), 4)
    y2 = torch.ones((N, Ho, Wo, K), dtype=dtype, device=DEV)
    ops.run([ops.rec_conv_fwd(xd, wp, y2, K, R, R, stride, pad, scale=sc.to(DEV), shift=sh.to(DEV), act=ops.ACT_SILU, accumulate=True)])
    ref2 = F.silu(ref * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)) + 1.0
    assert_close(from_dev_nhwc(y2), ref2, TOL[dtype] * 2, 'conv epilogue')

    if K % 8 or C % 8:
        return log_fwd, [], []
    # dgrad / wgrad against autograd
    dy = q(rnd((N, K, Ho, Wo), 5), dtype)
    xr = x.clone().requires_grad_(True)
    wr = wq.clone().requires_grad_(True)
    F.conv2d(xr, wr, None, stride, pad).backward(dy)
    dyd = to_dev_nhwc(dy, dtype, ld=K + 8, off=0)
    wpd = ops.pack_alloc(K, C, R, R, stride, pad, ops.PACK_DGRAD, dtype, DEV)
    dx = torch.full((N, H, W, C), 1.0, dtype=dtype, device=DEV)
    ops.run([ops.rec_pack(wdev, None, stride, pad, ops.PACK_DGRAD, wpd)])
    _lib.dispatch_log(reset=True)
    ops.run([ops.rec_conv_dgrad(dyd, wpd, dx, R, R, stride, pad, accumulate=True)])
    log_dgrad = _lib.dispatch_log(reset=True)
    assert_close(from_dev_nhwc(dx), xr.grad + 1.0, TOL[dtype] * 2, 'dgrad')

    ws = torch.empty(ops.wgrad_ws_bytes(N, H, W, C, K, R, R, stride, pad, dtype) // 4 + 1, dtype=torch.float32, device=DEV)
    ka = K // 2 if K >= 16 else K
    ga = torch.zeros((ka, C, R, R), dtype=torch.float32, device=DEV)
    gb = torch.full((K - ka, C, R, R), 2.0, dtype=torch.float32, device=DEV) if ka < K else None
    ops.run([ops.rec_conv_wgrad(xd, dyd, ga, gb, R, R, stride, pad, ws, accumulate=True)])
    log_wgrad = _lib.dispatch_log(reset=True)
    assert_close(ga.cpu(), wr.grad[:ka], TOL[dtype] * 3, 'wgrad a')
    if gb is not None:
        assert_close(gb.cpu(), wr.grad[ka:] + 2.0, TOL[dtype] * 3, 'wgrad b')
    return log_fwd, log_dgrad, log_wgrad


RES_CASES = [
    # (N, H, W, C, K, R, expected kernel family, options): the eval-mode Bottleneck shortcut — y = SiLU(scale * conv + shift) + res (layers.py:96) — in every conv kernel's epilogue
    (2, 16, 32, 64, 64, 3, 'conv3x3_c64', {}),
    (2, 24, 40, 128, 128, 3, 'conv3x3_c128', {}),
    (2, 16, 16, 128, 104, 3, 'conv3x3_c128', {}),                                      # ragged second channel half
    (3, 40, 40, 128, 128, 3, 'deep_256x128', {'HDY_NO_CONV3X3_C128': 1, 'HDY_DEEP_MIN_TILES': 1}),
    (3, 40, 40, 256, 256, 3, 'deep_256x256', {'HDY_DEEP_MIN_TILES': 1, 'HDY_DEEP_BN': 256}),
    (2, 20, 20, 96, 96, 3, 'igemm_128x128x2', {}),
    (2, 20, 20, 48, 48, 1, 'igemm_128x64x2', {}),
]


@pytest.mark.parametrize('case', RES_CASES, ids=[f'{c[6]}-{c[3]}x{c[4]}k{c[5]}' for c in RES_CASES])
def test_conv_eval_epilogue_with_residual(case):
    """The residual operand of the eval-mode convolution epilogue (a pitched slice of a wider buffer, like the plan's) on each kernel family, bf16, against
    torch fp32: out = SiLU(scale * conv(x, w) + shift) + res, rounded once."""
    from contextlib import ExitStack
    N, H, W, C, K, R, want, opts = case
    dt, pad = torch.bfloat16, R // 2
    x = q(rnd((N, C, H, W), 31), dt)
    w = rnd((K, C, R, R), 32, (3.0 / (C * R * R)) ** 0.5)
    res = q(rnd((N, K, H, W), 33), dt)
    sc, sh = rnd((K,), 34).abs() + 0.5, rnd((K,), 35)
    xd = to_dev_nhwc(x, dt)
    resd = to_dev_nhwc(res, dt, ld=K + 24, off=16)
    with ExitStack() as es:
        for k, v in opts.items():
            es.enter_context(_lib.option(k, v))
        wp = ops.pack_alloc(K, C, R, R, 1, pad, ops.PACK_FWD, dt, DEV)
        y = torch.zeros((N, H, W, K), dtype=dt, device=DEV)
        _lib.dispatch_log(reset=True)
        ops.run([ops.rec_pack(w.to(DEV), None, 1, pad, ops.PACK_FWD, wp),
                 ops.rec_conv_fwd(xd, wp, y, K, R, R, 1, pad, scale=sc.to(DEV), shift=sh.to(DEV), act=ops.ACT_SILU, res=resd)])
        log = _lib.dispatch_log(reset=True)
    assert want in log, log
    ref = F.silu(F.conv2d(x, q(w, dt), None, 1, pad) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)) + res
    assert_close(from_dev_nhwc(y), ref, TOL[dt] * 2, f'{want} epilogue with residual')


DEEP_CASES = [
    # N, H, W, C, K, R, stride, pad — shapes of the deep-pipelined kernel (C % 64 == 0, K >= 128), small enough for a CPU reference
    (2, 20, 20, 128, 128, 3, 1, 1),      # 4 row tiles, the last one 32 rows; every tile has border pixels
    (3, 40, 40, 128, 256, 3, 1, 1),      # tiles straddle images
    (1, 24, 56, 192, 128, 3, 1, 1),      # three channel blocks per tap
    (3, 40, 40, 256, 512, 1, 1, 0),      # 1x1: four K-tiles per output tile, the stream runs across tiles
    (1, 16, 16, 128, 256, 1, 1, 0),      # exactly one row tile
    (1, 16, 16, 192, 128, 1, 1, 0),      # one tile per workgroup and three K-tiles (with the cases above: K loops of 1, 2 and 3 in front of the statistics tail)
    (2, 48, 48, 128, 256, 3, 2, 1),      # stride-2 forward; its data gradient is the four-class walk on the deep pipeline (1 + 2 + 2 + 4 taps)
    (2, 40, 40, 256, 512, 3, 2, 1),      # ... with two 128-wide column tiles (one 256-wide)
    (1, 36, 44, 128, 128, 3, 2, 1),      # ... 396 class pixels: a ragged second row tile
    (2, 20, 20, 64, 384, 1, 1, 0),       # K = 1.5 column tiles of 256 / 3 of 128; one K-tile per output tile
    (1, 20, 20, 64, 136, 3, 1, 1),       # K not a multiple of 16: ragged last column tile
    (8, 96, 96, 64, 128, 1, 1, 0),       # 288 row tiles on 256 workgroups: some walk two tiles, one K-tile each
    (5, 96, 96, 64, 128, 3, 1, 1),       # 180 row tiles, nine K-tiles each
    (2, 37, 70, 192, 136, 3, 1, 1),      # three channel blocks per tap, K not a multiple of 16, ragged last row tile
    (70, 20, 20, 128, 256, 3, 1, 1),     # 110 row tiles of 18 K-tiles, two 128-wide column tiles: 220 tiles, every workgroup's stream crosses none / one boundary
]


@pytest.mark.parametrize('bn', [0, 128, 256])
@pytest.mark.parametrize('case', DEEP_CASES)
def test_deep_pipelined_conv(case, bn):
    """conv_deep.hip (256-row tiles, loads in flight across barriers) on shapes the plan would give to the generic kernel at test size:
    HDY_DEEP_MIN_TILES = 1 sends them to it; both column tiles (HDY_DEEP_BN); forward with BatchNorm sums, epilogue with scale / shift /
    SiLU / accumulate, stride-1 data gradient with accumulate — against torch fp32 on the CPU, and the dispatch log must name the kernel"""
    N, H, W, C, K, R, stride, pad = case
    with _lib.option('HDY_DEEP_MIN_TILES', 1), _lib.option('HDY_DEEP_BN', bn), _lib.option('HDY_DEEP_ALL', 1), _lib.option('HDY_DEEP_WALK', 1):
        log_fwd, log_dgrad, _ = conv_case(case, torch.bfloat16)
    assert log_fwd == ['deep_256x128'], log_fwd                  # statistics: 128-wide instances
    if stride == 1 and K % 64 == 0 and C >= 128:
        assert log_dgrad and log_dgrad[0].startswith('deep_256x'), log_dgrad
    if stride == 2 and K % 64 == 0 and C >= 128:
        assert log_dgrad and log_dgrad[0].startswith('deep_256x') and log_dgrad[0].endswith('_walk'), log_dgrad
        with _lib.option('HDY_DEEP_MIN_TILES', 1):                 # default (HDY_DEEP_WALK = 2): the deep pipeline from 256 gradient channels out, else the generic kernel's walk
            assert conv_case(case, torch.bfloat16)[1][0].startswith('deep_256x' if C >= 256 else 'igemm_')
        with _lib.option('HDY_DEEP_MIN_TILES', 1), _lib.option('HDY_DEEP_WALK', 0):
            assert conv_case(case, torch.bfloat16)[1][0].startswith('igemm_')
    # A/B: the same forward through the generic kernel agrees to the last bf16 rounding (both accumulate in fp32 over the same k order per tap)
    x = q(rnd((N, C, H, W), 11), torch.bfloat16)
    xd = to_dev_nhwc(x, torch.bfloat16)
    w = rnd((K, C, R, R), 12, (3.0 / (C * R * R)) ** 0.5).to(DEV)
    wp = ops.pack_alloc(K, C, R, R, stride, pad, ops.PACK_FWD, torch.bfloat16, DEV)
    Ho, Wo = ops.out_dim(H, R, stride, pad), ops.out_dim(W, R, stride, pad)
    outs = []
    for off in (0, 1):
        y = torch.zeros((N, Ho, Wo, K), dtype=torch.bfloat16, device=DEV)
        with _lib.option('HDY_DEEP_MIN_TILES', 1), _lib.option('HDY_DEEP_BN', bn), _lib.option('HDY_DEEP_ALL', 1), _lib.option('HDY_NO_DEEP', off):
            _lib.dispatch_log(reset=True)
            ops.run([ops.rec_pack(w, None, stride, pad, ops.PACK_FWD, wp), ops.rec_conv_fwd(xd, wp, y, K, R, R, stride, pad)])
            assert any(n.startswith('deep_') for n in _lib.dispatch_log()) == (off == 0)
        outs.append(y.float())
    d = (outs[0] - outs[1]).abs().max().item()
    assert d <= 2.0 ** -7 * outs[1].abs().max().item(), f'deep vs generic kernel differ by {d}'


@pytest.mark.parametrize('dtype', DTYPES)
def test_stacked_pack_matches_cat(dtype):
    C, Ka, Kb = 64, 32, 32
    wa, wb = rnd((Ka, C, 1, 1), 1).to(DEV), rnd((Kb, C, 1, 1), 2).to(DEV)
    p1 = ops.pack_alloc(Ka + Kb, C, 1, 1, 1, 0, ops.PACK_FWD, dtype, DEV)
    p2 = ops.pack_alloc(Ka + Kb, C, 1, 1, 1, 0, ops.PACK_FWD, dtype, DEV)
    ops.run([ops.rec_pack(wa, wb, 1, 0, ops.PACK_FWD, p1), ops.rec_pack(torch.cat([wa, wb]), None, 1, 0, ops.PACK_FWD, p2)])
    assert torch.equal(p1.float(), p2.float())


@pytest.mark.parametrize('K', [16, 32, 48, 64])
def test_stem_patch_kernel_bf16(K):
    """Shapes the patch-resident stem kernel takes (bf16, Ho % 16 == 0, Wo % 32 == 0): raw output + statistic slabs (train mode),
    scale/shift + SiLU (eval mode), against torch and against the generic kernel's path for an ineligible width."""
    if os.environ.get('HDY_NO_STEM_KERNEL'):
        pytest.skip('the patch-resident stem kernel is switched off')
    dtype = torch.bfloat16
    N, H, W = 3, 64, 192                      # 3 x 2 x 3 = 18 tiles of 16 x 32 outputs
    img = q(rnd((N, 3, H, W), 1).abs(), dtype)
    w = rnd((K, 3, 6, 6), 2, 0.2)
    wq = q(w, dtype)
    prep = torch.empty((N, H + 4, W + 4, 4), dtype=dtype, device=DEV)
    wp = ops.pack_alloc(K, 3, 6, 6, 2, 2, ops.PACK_STEM, dtype, DEV)
    Ho, Wo = H // 2, W // 2
    y = torch.empty((N, Ho, Wo, K), dtype=dtype, device=DEV)
    slabs = ops.stat_slabs(N, H, W, 3, K, 6, 6, 2, 2, dtype)
    assert slabs == 18                        # one per workgroup, not one per 128 pixels (72)
    stats = torch.full((slabs, 2, K), float('nan'), dtype=torch.float32, device=DEV)
    ops.run([ops.rec_stem_prep(img.to(DEV), prep), ops.rec_pack(w.to(DEV), None, 2, 2, ops.PACK_STEM, wp),
             ops.rec_conv_fwd(prep, wp, y, K, 6, 6, 2, 2, stats=stats, stem_hw=(H, W))])
    ref = F.conv2d(img, wq, None, 2, 2)
    assert_close(from_dev_nhwc(y), ref, TOL[dtype], 'stem fwd')
    s = stats.sum(0).cpu()
    assert_close(s[0], ref.sum((0, 2, 3)), 1e-3, 'stats sum')
    assert_close(s[1], (ref * ref).sum((0, 2, 3)), 1e-3, 'stats sumsq')
    scale, shift = rnd((K,), 5).abs() + 0.5, rnd((K,), 6)
    ops.run([ops.rec_conv_fwd(prep, wp, y, K, 6, 6, 2, 2, scale=scale.to(DEV), shift=shift.to(DEV), act=ops.ACT_SILU, stem_hw=(H, W))])
    assert_close(from_dev_nhwc(y), F.silu(ref * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)), TOL[dtype], 'stem fwd + bn + silu')


@pytest.mark.parametrize('dtype', DTYPES)
# output sizes that are multiples of 16 x 32 take the patch-resident weight-gradient kernel in bf16 (K = 16, 32, 48, 64; ragged tile counts)
@pytest.mark.parametrize('shape', [(2, 32, 48, 16), (2, 64, 128, 32), (3, 32, 64, 16), (1, 96, 64, 48), (2, 32, 192, 64)])
def test_stem_conv(shape, dtype):
    N, H, W, K = shape
    img = q(rnd((N, 3, H, W), 1).abs(), dtype)
    w = rnd((K, 3, 6, 6), 2, 0.2)
    wq = q(w, dtype)
    prep = torch.empty((N, H + 4, W + 4, 4), dtype=dtype, device=DEV)
    wp = ops.pack_alloc(K, 3, 6, 6, 2, 2, ops.PACK_STEM, dtype, DEV)
    Ho, Wo = H // 2, W // 2
    y = torch.empty((N, Ho, Wo, K), dtype=dtype, device=DEV)
    ops.run([ops.rec_stem_prep(img.to(DEV), prep), ops.rec_pack(w.to(DEV), None, 2, 2, ops.PACK_STEM, wp),
             ops.rec_conv_fwd(prep, wp, y, K, 6, 6, 2, 2, stem_hw=(H, W))])
    ref = F.conv2d(img, wq, None, 2, 2)
    assert_close(from_dev_nhwc(y), ref, TOL[dtype], 'stem fwd')
    dy = q(rnd((N, K, Ho, Wo), 3), dtype)
    wr = wq.clone().requires_grad_(True)
    F.conv2d(img, wr, None, 2, 2).backward(dy)
    ws = torch.empty(ops.wgrad_ws_bytes(N, H, W, 3, K, 6, 6, 2, 2, dtype, stem=True) // 4 + 1, dtype=torch.float32, device=DEV)
    g = torch.zeros((K, 3, 6, 6), dtype=torch.float32, device=DEV)
    ops.run([ops.rec_conv_wgrad(prep, to_dev_nhwc(dy, dtype), g, None, 6, 6, 2, 2, ws, stem_hw=(H, W))])
    assert_close(g.cpu(), wr.grad, TOL[dtype] * 3, 'stem wgrad')


@pytest.mark.parametrize('dtype', DTYPES)
# the last three shapes take the one-launch cooperative backward in bf16 (units in registers only / row tail / registers + LDS)
@pytest.mark.parametrize('shape', [(2, 20, 20, 64), (3, 9, 7, 48), (1, 4, 4, 512), (16, 40, 40, 64), (5, 37, 40, 256), (48, 40, 40, 256)])
def test_bn_silu_fwd_bwd(shape, dtype):
    N, H, W, K = shape
    M = N * H * W
    y = q(rnd((N, K, H, W), 1, 2.0) + 0.3, dtype)
    res = q(rnd((N, K, H, W), 2), dtype)
    gamma, beta = rnd((K,), 3).abs() + 0.5, rnd((K,), 4) * 0.3
    rm, rv = rnd((K,), 5) * 0.1, rnd((K,), 6).abs() + 0.5
    # reference
    yr = y.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    z_ref = F.silu(F.batch_norm(yr, rm_ref, rv_ref, gr, br, True, ops.BN_MOMENTUM, ops.BN_EPS)) + res
    dz = q(rnd((N, K, H, W), 7), dtype)
    z_ref.backward(dz)
    # device: per-tile statistics come from the conv kernel in production; emulate one slab here
    yd = to_dev_nhwc(y, dtype, ld=K + 8, off=8)
    stats = torch.stack([y.sum((0, 2, 3)), (y * y).sum((0, 2, 3))]).view(1, 2, K).to(DEV)
    dev = {k: v.to(DEV) for k, v in dict(gamma=gamma, beta=beta, rm=rm.clone(), rv=rv.clone()).items()}
    scale, shift, smean, sinv = (torch.empty(K, device=DEV) for _ in range(4))
    z = torch.empty((N, H, W, K), dtype=dtype, device=DEV)
    ops.run([ops.rec_bn_finalize(stats, 1, K, M, dev['gamma'], dev['beta'], dev['rm'], dev['rv'], scale, shift, smean, sinv),
             ops.rec_bn_act_fwd(yd, scale, shift, z, res=to_dev_nhwc(res, dtype))])
    assert_close(from_dev_nhwc(z), z_ref.detach(), TOL[dtype], 'bn+silu fwd')
    assert_close(dev['rm'].cpu(), rm_ref, 1e-5, 'running_mean')
    assert_close(dev['rv'].cpu(), rv_ref, 1e-5, 'running_var')
    dy = torch.empty((N, H, W, K), dtype=dtype, device=DEV)
    dg, db = torch.ones(K, device=DEV), torch.ones(K, device=DEV)
    ws = torch.empty(ops.bn_bwd_ws_floats(M, K), device=DEV)
    ops.run([ops.rec_bn_act_bwd(to_dev_nhwc(dz, dtype), yd, scale, shift, smean, sinv, dy, dg, db, ws, accumulate=True)])
    assert_close(from_dev_nhwc(dy), yr.grad, TOL[dtype] * 3, 'bn+silu bwd dy')
    assert_close(dg.cpu() - 1, gr.grad, TOL[dtype] * 3 if dtype == torch.float32 else 2e-3, 'dgamma')
    assert_close(db.cpu() - 1, br.grad, TOL[dtype] * 3 if dtype == torch.float32 else 2e-3, 'dbeta')
    # eval coefficients == fuse_conv_and_bn folding
    es, eh = torch.empty(K, device=DEV), torch.empty(K, device=DEV)
    ops.run([ops.rec_bn_eval_coeffs(dev['gamma'], dev['beta'], dev['rm'], dev['rv'], es, eh)])
    s_ref = gamma / torch.sqrt(dev['rv'].cpu() + ops.BN_EPS)
    assert_close(es.cpu(), s_ref, 1e-6, 'eval scale')
    assert_close(eh.cpu(), beta - dev['rm'].cpu() * s_ref, 1e-6, 'eval shift')
    # add_inplace
    a = to_dev_nhwc(y, dtype)
    b = to_dev_nhwc(res, dtype)
    ops.run([ops.rec_add_inplace(a, b)])
    assert_close(from_dev_nhwc(a), q(y + res, dtype), 1e-6 if dtype == torch.float32 else 8e-3, 'add_inplace')


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape', [(2, 20, 20, 32, 32), (3, 9, 7, 16, 48), (4, 40, 40, 64, 64)])
def test_bn_pair_equals_two_single_calls(shape, dtype):
    """The pair entry points (one K = Ka + Kb wide pass, parameters / outputs / gradient sources split at Ka) against the single-module
    calls on the two channel ranges: identical arithmetic per channel (forward exactly; backward up to the summation order)."""
    N, H, W, Ka, Kb = shape
    K, M = Ka + Kb, N * H * W
    y = to_dev_nhwc(q(rnd((N, K, H, W), 1, 2.0) + 0.3, dtype), dtype)
    stats = torch.stack([from_dev_nhwc(y).sum((0, 2, 3)), (from_dev_nhwc(y) ** 2).sum((0, 2, 3))]).view(1, 2, K).to(DEV)
    par = {k: (rnd((K,), i + 3).abs() + 0.5).to(DEV) for i, k in enumerate(('gamma', 'beta', 'rm', 'rv'))}
    dza = to_dev_nhwc(q(rnd((N, Ka, H, W), 8), dtype), dtype, ld=Ka + 8, off=8)
    dzb = to_dev_nhwc(q(rnd((N, Kb, H, W), 9), dtype), dtype, ld=Kb + 16, off=0)

    def run(pair):
        p = {k: v.clone() for k, v in par.items()}
        sl = lambda t, a, b: t[a:b]
        scale, shift, mean, inv = (torch.empty(K, device=DEV) for _ in range(4))
        za = torch.zeros((N, H, W, Ka + 8), dtype=dtype, device=DEV)[..., 8:]
        zb = torch.zeros((N, H, W, Kb), dtype=dtype, device=DEV)
        dy = torch.zeros((N, H, W, K), dtype=dtype, device=DEV)
        dg, db = torch.zeros(K, device=DEV), torch.zeros(K, device=DEV)
        ws = torch.empty(ops.bn_bwd_ws_floats(M, K), device=DEV)
        if pair:
            bn = lambda a, b: tuple(sl(p[k], a, b) for k in ('gamma', 'beta', 'rm', 'rv'))
            ops.run([ops.rec_bn_finalize_pair(stats, 1, K, Ka, M, bn(0, Ka), bn(Ka, K), scale, shift, mean, inv),
                     ops.rec_bn_act_fwd_pair(y, scale, shift, za, zb),
                     ops.rec_bn_act_bwd_pair(dza, dzb, y, scale, shift, mean, inv, dy, dg[:Ka], db[:Ka], dg[Ka:], db[Ka:], ws)])
        else:
            for a, b, z, dz in ((0, Ka, za, dza), (Ka, K, zb, dzb)):
                ops.run([ops.rec_bn_finalize(stats[:, :, a:], 1, b - a, M, p['gamma'][a:b], p['beta'][a:b], p['rm'][a:b], p['rv'][a:b], scale[a:], shift[a:],
                                             mean[a:], inv[a:], stats_ld=K),
                         ops.rec_bn_act_fwd(y[..., a:b], scale[a:b], shift[a:b], z),
                         ops.rec_bn_act_bwd(dz, y[..., a:b], scale[a:b], shift[a:b], mean[a:b], inv[a:b], dy[..., a:b], dg[a:b], db[a:b], ws)])
        torch.cuda.synchronize()
        return [t.float().cpu() for t in (za, zb, dy, dg, db, p['rm'], p['rv'], scale, shift)]

    for name, a, b in zip(('z_a', 'z_b', 'dy', 'dgamma', 'dbeta', 'running_mean', 'running_var', 'scale', 'shift'), run(True), run(False)):
        if name in ('dy', 'dgamma', 'dbeta'):        # the K-wide pass partitions the rows differently among lanes: fp32 sums in another order
            assert_close(a, b, 2e-5 if dtype == torch.float32 else 8e-3, name)
        else:
            assert torch.equal(a, b), name


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape', [(2, 20, 20, 32), (1, 5, 7, 8), (1, 32, 32, 16), (1, 40, 40, 8), (2, 32, 32, 64), (1, 40, 40, 32)])
def test_sppf_pool(shape, dtype):
    N, H, W, C = shape
    x = q(rnd((N, C, H, W), 1), dtype)
    xr = x.clone().requires_grad_(True)
    y1 = F.max_pool2d(xr, 5, 1, 2)
    y2 = F.max_pool2d(y1, 5, 1, 2)
    y3 = F.max_pool2d(y2, 5, 1, 2)
    cat = torch.cat([xr, y1, y2, y3], 1)
    g = q(rnd(cat.shape, 2), dtype)
    cat.backward(g)
    buf = torch.zeros((N, H, W, 4 * C), dtype=dtype, device=DEV)
    sl = [buf[..., i * C:(i + 1) * C] for i in range(4)]
    sl[0].copy_(x.permute(0, 2, 3, 1).to(dtype))
    idx = [torch.empty((N, H, W, C), dtype=torch.uint8, device=DEV) for _ in range(3)]
    ops.run([ops.rec_sppf_pool_fwd(sl[0], sl[1], sl[2], sl[3], idx)])
    assert torch.equal(from_dev_nhwc(buf), cat.detach()), 'sppf forward must be exact'
    # inference form (no window positions): row maximum then column maximum, same values; a NaN must poison its whole window
    buf2 = torch.zeros_like(buf)
    sl2 = [buf2[..., i * C:(i + 1) * C] for i in range(4)]
    sl2[0].copy_(sl[0])
    ops.run([ops.rec_sppf_pool_fwd(sl2[0], sl2[1], sl2[2], sl2[3], None)])
    assert torch.equal(buf2, buf), 'sppf forward without argmax must equal the training form'
    xn = x.clone()
    xn[0, 0, H // 2, W // 2] = float('nan')
    sl2[0].copy_(xn.permute(0, 2, 3, 1).to(dtype))
    ops.run([ops.rec_sppf_pool_fwd(sl2[0], sl2[1], sl2[2], sl2[3], None)])
    want = torch.cat([xn] + [F.max_pool2d(xn, k, 1, k // 2) for k in (5, 9, 13)], 1)
    assert torch.equal(torch.isnan(from_dev_nhwc(buf2)), torch.isnan(want))
    gd = to_dev_nhwc(g, dtype)
    gs = [gd[..., i * C:(i + 1) * C] for i in range(4)]
    dx = torch.empty((N, H, W, C), dtype=dtype, device=DEV)
    ops.run([ops.rec_sppf_pool_bwd(gs[0], gs[1], gs[2], gs[3], idx, dx)])
    assert_close(from_dev_nhwc(dx), xr.grad, TOL[dtype], 'sppf backward')
    # many equal values per window (four levels): the gradient must go to the FIRST maximum in row-major window order, as ATen's does
    xt = q(torch.round(x * 1.5) / 1.5, dtype).clone().requires_grad_(True)
    y1 = F.max_pool2d(xt, 5, 1, 2)
    y2 = F.max_pool2d(y1, 5, 1, 2)
    y3 = F.max_pool2d(y2, 5, 1, 2)
    cat = torch.cat([xt, y1, y2, y3], 1)
    cat.backward(g)
    sl[0].copy_(xt.detach().permute(0, 2, 3, 1).to(dtype))
    ops.run([ops.rec_sppf_pool_fwd(sl[0], sl[1], sl[2], sl[3], idx)])
    assert torch.equal(from_dev_nhwc(buf), cat.detach()), 'sppf forward with ties'
    ops.run([ops.rec_sppf_pool_bwd(gs[0], gs[1], gs[2], gs[3], idx, dx)])
    assert_close(from_dev_nhwc(dx), xt.grad, TOL[dtype], 'sppf backward with ties')


@pytest.mark.parametrize('dtype', DTYPES)
def test_upsample_and_layout(dtype):
    N, H, W, C = 2, 5, 6, 16
    x = q(rnd((N, C, H, W), 1), dtype)
    xd = to_dev_nhwc(x, dtype)
    y = torch.empty((N, 2 * H, 2 * W, C + 8), dtype=dtype, device=DEV)[..., :C]
    ops.run([ops.rec_upsample_fwd(xd, y)])
    assert torch.equal(from_dev_nhwc(y), F.interpolate(x, scale_factor=2.0, mode='nearest'))
    g = q(rnd((N, C, 2 * H, 2 * W), 2), dtype)
    dx = torch.ones((N, H, W, C), dtype=dtype, device=DEV)
    ops.run([ops.rec_upsample_bwd(to_dev_nhwc(g, dtype), dx, accumulate=True)])
    ref = g.view(N, C, H, 2, W, 2).sum((3, 5)) + 1
    assert_close(from_dev_nhwc(dx), ref, TOL[dtype], 'upsample bwd')
    src = rnd((2, 37, 9, 11), 3)
    dst = torch.empty((2, 9, 11, 40), dtype=dtype, device=DEV)[..., :37]
    ops.run([ops.rec_nchw_to_nhwc(src.to(DEV), dst)])
    assert torch.equal(from_dev_nhwc(dst), q(src, dtype))


def test_decode_against_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, 'decode.npz'))
    strides = [8.0, 16.0, 32.0]
    for tag in ('default', 'odd'):
        dets = [torch.from_numpy(g[f'{tag}_det_{i}']).to(DEV) for i in range(3)]
        B, na, no = dets[0].shape[0], dets[0].shape[1], dets[0].shape[4]
        rows = sum(d.shape[1] * d.shape[2] * d.shape[3] for d in dets)
        out = torch.zeros((B, rows, no + 1), device=DEV)
        off = 0
        for l, d in enumerate(dets):
            ops.decode_level(d, synth.ANCHORS_P5[l], strides[l], out, off, l)
            n = d.shape[1] * d.shape[2] * d.shape[3]
            ref = torch.from_numpy(g[f'{tag}_pred_{l}']).reshape(B, n, no)
            got = out[:, off:off + n].cpu()
            # 1e-4 relative (north star) with an absolute floor of 1e-4 px on coordinates
            np.testing.assert_allclose(got[..., :no].numpy(), ref.numpy(), rtol=1e-4, atol=1e-4)
            assert (got[..., no] == float(l)).all()
            off += n
        # NHWC-strided input view (what the det conv writes) gives the same rows
        d = dets[0]
        nhwc = d.permute(0, 2, 3, 1, 4).contiguous()            # (B, ny, nx, na, no)
        out2 = torch.zeros((B, d.shape[1] * d.shape[2] * d.shape[3], no + 1), device=DEV)
        ops.decode_level(nhwc.permute(0, 3, 1, 2, 4), synth.ANCHORS_P5[0], 8.0, out2, 0, 0)
        assert torch.equal(out2, out[:, :out2.shape[1]])
        # the plan's layout: pixel-major logits with the channel pitch padded to a multiple of 4 floats -> the tiled kernel; every level,
        # written at its row offset, must give the very same bits as the per-candidate kernel
        out3 = torch.zeros_like(out)
        off = 0
        for l, d in enumerate(dets):
            Bn, na_, ny, nx, no_ = d.shape
            ld = (na_ * no_ + 7) // 8 * 8
            buf = torch.full((Bn, ny, nx, ld), 9.0, device=DEV)
            buf[..., :na_ * no_] = d.permute(0, 2, 3, 1, 4).reshape(Bn, ny, nx, na_ * no_)
            view = buf[..., :na_ * no_].view(Bn, ny, nx, na_, no_).permute(0, 3, 1, 2, 4)
            ops.decode_level(view, synth.ANCHORS_P5[l], strides[l], out3, off, l)
            off += na_ * ny * nx
        assert torch.equal(out3, out)


def _check_nms(preds, nc, conf, iou, max_det, class_aware=False):
    from oracle import nms_ref
    res = ops.nms_batched(preds.to(DEV), nc, conf, iou, max_det, class_aware=class_aware)
    keep, nk, cls = nms_ref.nms_batched_c(preds.numpy(), nc, conf, iou, max_det, class_aware=class_aware)
    got_keep, got_nk = res['keep'].cpu().numpy(), res['n_keep'].cpu().numpy()
    assert np.array_equal(got_nk, nk), (got_nk, nk)
    assert np.array_equal(got_keep, keep), 'kept indices / order differ from the oracle'
    for b in range(preds.shape[0]):
        n = nk[b]
        rows = preds[b, keep[b, :n]].numpy()
        hw = rows[:, 2:4] / np.float32(2)
        ref_boxes = np.concatenate([rows[:, :2] - hw, rows[:, :2] + hw], 1)
        assert np.array_equal(res['boxes'][b, :n].cpu().numpy(), ref_boxes)
        assert np.array_equal(res['scores'][b, :n].cpu().numpy(), rows[:, 4:5 + nc])
        if preds.shape[2] > 5 + nc:
            assert np.array_equal(res['extra'][b, :n].cpu().numpy(), rows[:, 5 + nc:])
        if class_aware:
            assert np.array_equal(res['cls'][b, :n].cpu().numpy(), cls[b, :n])
    return nk


@pytest.mark.parametrize('m,extra,max_det', [(64, 100, 300), (700, 3000, 300), (3000, 22200, 300), (3000, 2000, 2000),
                                             (9000, 1000, 300), (9000, 1000, 4096)])
def test_nms_matches_oracle_bit_exact(m, extra, max_det):
    preds = synth.synth_nms_preds(3, m, nc=8, extra=extra)
    nk = _check_nms(preds, 8, 0.15, 0.45, max_det)
    assert (nk > 0).all()


def test_nms_class_aware_and_edges():
    preds = synth.synth_nms_preds(2, 1500, nc=4, extra=500)
    _check_nms(preds, 4, 0.15, 0.45, 300, class_aware=True)
    # docstring example of utils_general.py:303-307 (+ level column)
    ex = torch.tensor([[[25, 25, 50, 50, 0.9, 0.5, 0.6, 0.0], [26, 26, 50, 50, 0.8, 0.9, 0.1, 1.0]]])
    assert _check_nms(ex, 2, 0.25, 0.45, 300).tolist() == [1]
    assert _check_nms(ex, 2, 0.25, 0.45, 300, class_aware=True).tolist() == [2]
    # ties, all-suppressed, nothing above conf, single row, N not a multiple of the block
    t = torch.zeros((1, 5, 7))
    t[0, :, :4] = torch.tensor([10., 10., 8., 8.])
    t[0, :, 4] = 0.5
    assert _check_nms(t, 1, 0.15, 0.45, 10).tolist() == [1]
    assert _check_nms(torch.zeros((2, 1030, 7)), 1, 0.15, 0.45, 10).tolist() == [0, 0]
    one = torch.tensor([[[5., 5., 4., 4., 0.9, 0.3, 2.0]]])
    assert _check_nms(one, 1, 0.15, 0.45, 1).tolist() == [1]
    # heavy ties on score with distinct boxes: order must be by row index
    g = torch.Generator().manual_seed(5)
    p = torch.zeros((1, 2000, 7))
    p[0, :, 0:2] = torch.rand((2000, 2), generator=g) * 600
    p[0, :, 2:4] = 10 + torch.rand((2000, 2), generator=g) * 20
    p[0, :, 4] = (torch.randint(2, 10, (2000,), generator=g).float() / 10)
    p[0, :, 5] = 0.5
    _check_nms(p, 1, 0.15, 0.45, 300)


@pytest.mark.parametrize('max_det', [5000, 20000])
def test_nms_per_image_with_max_det_beyond_the_lds_kept_list(max_det):
    """the reference slices `[:max_det]` for any value (utils_general.py:342); beyond 4096 kept boxes the kernel's list
    lives in its workspace.  M = 16 384 survivors, sparse boxes so that most of them are kept: bit-exact order against oracle/nms_ref.c, for
    two tiles in one launch (the second one dense: its list stays short), and the result of a max_det that is not reached equals the full one."""
    g = torch.Generator().manual_seed(max_det)
    N = 16384
    preds = torch.zeros((2, N, 13))
    preds[0, :, 0:2] = torch.rand((N, 2), generator=g) * 6000           # sparse: nearly every box survives
    preds[1, :, 0:2] = torch.rand((N, 2), generator=g) * 120            # dense: a short kept list in the same launch
    preds[:, :, 2:4] = 8 + torch.rand((2, N, 2), generator=g) * 24
    preds[:, :, 4] = 0.2 + torch.rand((2, N), generator=g) * 0.8
    preds[0, ::13, 4] = preds[0, 5, 4]                                  # ties: lower row first
    preds[:, :, 5:] = torch.rand((2, N, 8), generator=g)
    nk = _check_nms(preds, 8, 0.15, 0.45, max_det)
    assert nk[0] > 4096 and nk[1] < 4096
    if max_det == 20000:
        assert nk[0] < max_det                                           # every survivor returned, not a truncated list


def test_errors_are_loud():
    with pytest.raises(_lib.HdyError):
        ops.nms_batched(torch.zeros((1, 4, 7), device=DEV), 1, 0.15, 0.45, 0)         # max_det must be positive (any positive value is served)
    with pytest.raises(_lib.HdyError):
        ops.require_gpu(torch.zeros(1))
    x = torch.zeros((1, 4, 4, 12), dtype=torch.bfloat16, device=DEV)                   # C not a multiple of 8
    y = torch.zeros((1, 4, 4, 8), dtype=torch.bfloat16, device=DEV)
    wp = torch.zeros(4096, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(_lib.HdyError):
        ops.run([ops.rec_conv_fwd(x, wp, y, 8, 1, 1, 1, 0)])


# ---------------------------------------------------------------------------------------------- fused 1x1 backward (conv1x1_bwd.hip)
@pytest.mark.parametrize('K,M,pair,acc', [(32, 128 * 3 + 37, False, False), (64, 128 * 5 + 1, True, True), (64, 90, False, True),
                                          (128, 128 * 4 + 77, True, False), (128, 128 * 150 + 5, False, True), (64, 128 * 180, True, False),
                                          # more than 512 tiles: workgroups walk 2-3 tiles (requests in flight across tiles), partial last tile
                                          (64, (7, 18839), True, True), (32, (5, 17921), False, False), (64, (4, 19200), False, False),
                                          (32, (10, 19210), True, True), (128, (7, 10057), True, True),
                                          # round 6: the 96-channel instance (yolov5m: 12 chunks per row, 192 of 256 threads in the element-wise and store phases)
                                          (96, 64 * 3 + 37, False, False), (96, 64 * 5 + 1, True, True), (96, 50, False, True), (96, 64 * 150 + 5, False, True),
                                          (96, (7, 10057), True, True), (96, (4, 19200), False, False)])
def test_fused_1x1_backward_matches_reference_and_the_three_launch_path(K, M, pair, acc):
    """hdy_conv1x1_bwd_fused (BatchNorm/SiLU backward apply + wgrad + dgrad in one pass) against (a) plain torch fp32 on the bf16-rounded
    operands and (b) the three-launch path (hdy_bn_act_bwd -> dy, hdy_conv_wgrad, hdy_conv_dgrad) it replaces: same dy bits, so dx / dW
    differ by accumulation order only.  Operands are channel slices of wider buffers; the pair case splits dz over two tensors."""
    dt = torch.bfloat16
    C = K
    Nb, Wd = M if isinstance(M, tuple) else (1, M)                         # pixels as Nb rows of Wd (the generic kernels address 16-bit image sides)
    M = Nb * Wd
    flat = lambda t: t.permute(0, 2, 3, 1).reshape(M, t.shape[1])          # (Nb, ch, 1, Wd) -> (M, ch)
    dz = rnd((Nb, K, 1, Wd), 1)
    y = rnd((Nb, K, 1, Wd), 2, 2.0) + rnd((1, K, 1, 1), 3)                # per-channel offsets: non-trivial mean / invstd
    x = rnd((Nb, C, 1, Wd), 4)
    w = rnd((K, C, 1, 1), 5, 0.3)
    dx0 = rnd((Nb, C, 1, Wd), 6, 0.5)
    gamma, beta = rnd((K,), 7) + 1.5, rnd((K,), 8, 0.3)
    Ka = K // 2 if pair else K
    dz_a = to_dev_nhwc(dz[:, :Ka], dt, ld=Ka + 16, off=8)
    dz_b = to_dev_nhwc(dz[:, Ka:], dt, ld=K + 8, off=0) if pair else None
    yd = to_dev_nhwc(y, dt)
    xd = to_dev_nhwc(x, dt, ld=C + 24, off=16)
    dzq, yq, xq, wq = flat(q(dz, dt)), flat(q(y, dt)), flat(q(x, dt)), q(w, dt)[:, :, 0, 0]                          # (M, K) (M, K) (M, C) (K, C)
    # BatchNorm coefficients as the forward pass would have left them (batch statistics of y)
    mean, var = yq.mean(0), yq.var(0, unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-3)
    scale, shift = gamma * invstd, beta - mean * gamma * invstd
    u = yq * scale + shift
    sg = torch.sigmoid(u)
    du = dzq * (sg * (1 + u * (1 - sg)))
    xh = (yq - mean) * invstd
    c1, c2 = du.mean(0), (du * xh).mean(0)
    dy = q(scale * (du - c1 - xh * c2), dt)
    ref_dx = dy @ wq + (flat(q(dx0, dt)) if acc else 0.0)
    ref_dw = dy.T @ xq
    dev = lambda t: t.float().contiguous().to(DEV)
    sc_d, sh_d, mu_d, is_d = dev(scale), dev(shift), dev(mean), dev(invstd)
    ws = torch.empty(ops.bn_bwd_ws_floats(M, K), dtype=torch.float32, device=DEV)
    dgam, dbet = torch.zeros(K, device=DEV), torch.zeros(K, device=DEV)
    wpd = ops.pack_alloc(K, C, 1, 1, 1, 0, ops.PACK_DGRAD, dt, DEV)
    ops.run([ops.rec_pack(w.to(DEV), None, 1, 0, ops.PACK_DGRAD, wpd)])

    def stats_rec(dy_out):
        if pair:
            return ops.rec_bn_act_bwd_pair(dz_a, dz_b, yd, sc_d, sh_d, mu_d, is_d, dy_out, dgam[:Ka], dbet[:Ka], dgam[Ka:], dbet[Ka:], ws)
        return ops.rec_bn_act_bwd(dz_a, yd, sc_d, sh_d, mu_d, is_d, dy_out, dgam, dbet, ws)

    # (b) three launches
    dy_d = torch.empty((Nb, 1, Wd, K), dtype=dt, device=DEV)
    dx_b = to_dev_nhwc(dx0, dt, ld=C + 8, off=8)
    gw_b = torch.zeros((K, C, 1, 1), device=DEV)
    wgws = torch.empty(ops.wgrad_ws_bytes(1, 1, M, C, K, 1, 1, 1, 0, dt) // 4 + 16, dtype=torch.float32, device=DEV)
    ops.run([stats_rec(dy_d), ops.rec_conv_wgrad(xd, dy_d, gw_b[:Ka], gw_b[Ka:] if pair else None, 1, 1, 1, 0, wgws),
             ops.rec_conv_dgrad(dy_d, wpd, dx_b, 1, 1, 1, 0, accumulate=acc)])
    # (a) fused
    dx_f = to_dev_nhwc(dx0, dt, ld=C + 8, off=8)
    gw_f = torch.zeros((K, C, 1, 1), device=DEV)
    f1ws = torch.empty(ops.fused_1x1_ws_bytes(M, C, K) // 4 + 16, dtype=torch.float32, device=DEV)
    c1_d, c2_d = ops.bn_bwd_coeffs(ws, M, K)
    ops.run([stats_rec(None),
             ops.rec_conv1x1_bwd_fused(dz_a, dz_b, yd, sc_d, sh_d, mu_d, is_d, c1_d, c2_d, xd, wpd, dx_f, gw_f[:Ka], gw_f[Ka:] if pair else None, f1ws,
                                       accumulate_dx=acc)])
    torch.cuda.synchronize()
    assert_close(c1_d.cpu(), c1, 2e-3, 'c1')
    got_dx, got_dw = dx_f.float().cpu().reshape(M, C), gw_f.cpu()[:, :, 0, 0]
    assert_close(got_dx, ref_dx, 1.5e-2, 'dx vs torch')
    assert_close(got_dw, ref_dw, 1.5e-2, 'dW vs torch')
    assert_close(got_dx, dx_b.float().cpu().reshape(M, C), 8e-3, 'dx vs three launches')         # one bf16 rounding of the output apart at most
    assert_close(got_dw, gw_b.cpu()[:, :, 0, 0], 2e-4, 'dW vs three launches', elementwise=False)   # two bf16-operand paths: the three launches round dy to bf16 in between, the fused kernel does not — elements that cancel differ by more than a summation order
    # dgrad only / wgrad only
    dx_o = to_dev_nhwc(dx0, dt, ld=C + 8, off=8)
    gw_o = torch.zeros((K, C, 1, 1), device=DEV)
    ops.run([ops.rec_conv1x1_bwd_fused(dz_a, dz_b, yd, sc_d, sh_d, mu_d, is_d, c1_d, c2_d, xd, wpd, dx_o, None, None, f1ws, accumulate_dx=acc),
             ops.rec_conv1x1_bwd_fused(dz_a, dz_b, yd, sc_d, sh_d, mu_d, is_d, c1_d, c2_d, xd, None, None, gw_o[:Ka], gw_o[Ka:] if pair else None, f1ws)])
    torch.cuda.synchronize()
    assert torch.equal(dx_o, dx_f) and torch.equal(gw_o, gw_f)
    assert xd._base[..., :16].float().eq(7.0).all()         # poison outside the slices untouched
    assert dx_f._base[..., :8].float().eq(7.0).all()


@pytest.mark.parametrize('case', [(2, 24, 20, 64, 32, 1, 1, 0, False), (3, 16, 16, 32, 64, 3, 1, 1, True), (2, 16, 24, 64, 128, 3, 2, 1, False),
                                  (2, 40, 40, 64, 128, 3, 1, 1, False), (5, 128, 128, 32, 64, 3, 2, 1, True)])
def test_dgrad_serves_batchnorm_backward_statistics(case):
    """hdy_conv_dgrad_stats: the data-gradient launch that completes dx also leaves, for two channel ranges of dx that are the output
    gradients of two Conv+BN+SiLU units, per-workgroup slabs of (SUM du, SUM du*xhat); hdy_bn_bwd_finalize_slabs reduces them to the same
    dgamma / dbeta / c1 / c2 as the unit's own reduce pass (hdy_bn_act_bwd) on the finished dx."""
    N, H, W, C, K, R, stride, pad, acc = case
    dt = torch.bfloat16
    Ho, Wo = ops.out_dim(H, R, stride, pad), ops.out_dim(W, R, stride, pad)
    dy = to_dev_nhwc(rnd((N, K, Ho, Wo), 1), dt)
    w = rnd((K, C, R, R), 2, 0.2)
    wpd = ops.pack_alloc(K, C, R, R, stride, pad, ops.PACK_DGRAD, dt, DEV)
    ops.run([ops.rec_pack(w.to(DEV), None, stride, pad, ops.PACK_DGRAD, wpd)])
    dx0 = rnd((N, C, H, W), 3, 0.5)
    nslabs = ops.conv_dgrad_stat_slabs(N, H, W, C, K, R, R, stride, pad, dt)
    assert nslabs > 0 and ops.conv_dgrad_stat_slabs(N, H, W, 128, K, R, R, stride, pad, dt) == 0     # 128-wide gradients are not served
    Ca = C // 2                                                      # two units: channels [0, Ca) and [Ca, C), their raw outputs in separate buffers
    units = []
    for i, (c0, c1) in enumerate([(0, Ca), (Ca, C)]):
        y = to_dev_nhwc(rnd((N, c1 - c0, H, W), 10 + i, 2.0) + rnd((1, c1 - c0, 1, 1), 20 + i), dt, ld=c1 - c0 + 8, off=8 * i)
        yq = y.float()
        mean, var = yq.mean((0, 1, 2)), yq.var((0, 1, 2), unbiased=False)
        invstd = 1.0 / torch.sqrt(var + 1e-3)
        gamma, beta = (rnd((c1 - c0,), 30 + i) + 1.5).to(DEV), rnd((c1 - c0,), 40 + i, 0.3).to(DEV)
        scale, shift = (gamma * invstd).contiguous(), (beta - mean * gamma * invstd).contiguous()
        slabs = torch.zeros((nslabs, 2, c1 - c0), device=DEV)
        units.append(dict(y=y, scale=scale, shift=shift, mean=mean.contiguous(), invstd=invstd.contiguous(), slabs=slabs, c0=c0, c1=c1))
    reqs = [ops.StatRequest(u['y'], u['scale'], u['shift'], u['slabs'], u['c0'], ops.ACT_SILU) for u in units]
    dx = to_dev_nhwc(dx0, dt, ld=C + 16, off=8)
    dx_plain = to_dev_nhwc(dx0, dt, ld=C + 16, off=8)
    ops.run([ops.rec_conv_dgrad(dy, wpd, dx, R, R, stride, pad, accumulate=acc, stats=reqs),
             ops.rec_conv_dgrad(dy, wpd, dx_plain, R, R, stride, pad, accumulate=acc)])
    torch.cuda.synchronize()
    assert_close(dx.float().cpu(), dx_plain.float().cpu(), 8e-3, 'dx with / without statistics (generic vs filter-resident kernel)')
    M = N * H * W
    for u in units:
        Kc = u['c1'] - u['c0']
        dg, db, c1, c2 = (torch.zeros(Kc, device=DEV) for _ in range(4))
        ops.run([ops.rec_bn_bwd_finalize_slabs(u['slabs'], M, u['mean'], u['invstd'], dg, db, c1, c2)])
        # the unit's own reduce pass over the finished gradient
        ws = torch.empty(ops.bn_bwd_ws_floats(M, Kc), dtype=torch.float32, device=DEV)
        dg2, db2 = torch.zeros(Kc, device=DEV), torch.zeros(Kc, device=DEV)
        ops.run([ops.rec_bn_act_bwd(dx[..., u['c0']:u['c1']], u['y'], u['scale'], u['shift'], u['mean'], u['invstd'], None, dg2, db2, ws)])
        r1, r2 = ops.bn_bwd_coeffs(ws, M, Kc)
        torch.cuda.synchronize()
        for got, ref, what in ((dg, dg2, 'dgamma'), (db, db2, 'dbeta'), (c1, r1, 'c1'), (c2, r2, 'c2')):
            assert_close(got.cpu(), ref.cpu(), 1e-4, what)       # SUM du*xhat as invstd*(SUM du*y - mean*SUM du): mild cancellation
    assert dx._base[..., :8].float().eq(7.0).all()


@pytest.mark.parametrize('K,M,acc', [(64, 128 * 3 + 50, False), (32, 128 * 1100 + 33, True), (64, 128 * 1100 + 33, False), (64, 128 * 700, True)])
def test_fused_1x1_backward_serves_batchnorm_backward_statistics(K, M, acc):
    """hdy_conv1x1_bwd_fused_stats: the fused launch that completes dx also leaves, for the two Conv+BN+SiLU units whose output gradient dx is,
    per-workgroup slabs of (SUM du, SUM du*y); dx itself is bit-identical to the launch without requests, and hdy_bn_bwd_finalize_slabs gives the
    dgamma / dbeta / c1 / c2 of each unit's own reduce pass over the finished dx.  The long cases walk several tiles per workgroup."""
    dt = torch.bfloat16
    C = K
    dz = to_dev_nhwc(rnd((1, K, 1, M), 1), dt)
    yd = to_dev_nhwc(rnd((1, K, 1, M), 2, 2.0) + rnd((1, K, 1, 1), 3), dt)
    xd = to_dev_nhwc(rnd((1, C, 1, M), 4), dt)
    w = rnd((K, C, 1, 1), 5, 0.3)
    dx0 = rnd((1, C, 1, M), 6, 0.5)
    yq = yd.float()[0, 0]
    mean, var = yq.mean(0), yq.var(0, unbiased=False)
    invstd = (1.0 / torch.sqrt(var + 1e-3)).contiguous()
    gamma, beta = (rnd((K,), 7) + 1.5).to(DEV), rnd((K,), 8, 0.3).to(DEV)
    scale, shift = (gamma * invstd).contiguous(), (beta - mean * gamma * invstd).contiguous()
    ws = torch.empty(ops.bn_bwd_ws_floats(M, K), dtype=torch.float32, device=DEV)
    dgam, dbet = torch.zeros(K, device=DEV), torch.zeros(K, device=DEV)
    wpd = ops.pack_alloc(K, C, 1, 1, 1, 0, ops.PACK_DGRAD, dt, DEV)
    ops.run([ops.rec_pack(w.to(DEV), None, 1, 0, ops.PACK_DGRAD, wpd), ops.rec_bn_act_bwd(dz, yd, scale, shift, mean.contiguous(), invstd, None, dgam, dbet, ws)])
    c1_d, c2_d = ops.bn_bwd_coeffs(ws, M, K)
    nslabs = ops.fused_1x1_stat_slabs(M, C, K, dt)
    assert nslabs > 0 and ops.fused_1x1_stat_slabs(M, 128, 128, dt) == 0
    Ca = C // 2
    units = []
    for i, (c0, c1) in enumerate([(0, Ca), (Ca, C)]):
        y = to_dev_nhwc(rnd((1, c1 - c0, 1, M), 10 + i, 2.0) + rnd((1, c1 - c0, 1, 1), 20 + i), dt, ld=c1 - c0 + 8, off=8 * i)
        uq = y.float()
        um, uv = uq.mean((0, 1, 2)), uq.var((0, 1, 2), unbiased=False)
        uis = (1.0 / torch.sqrt(uv + 1e-3)).contiguous()
        g2, b2 = (rnd((c1 - c0,), 30 + i) + 1.5).to(DEV), rnd((c1 - c0,), 40 + i, 0.3).to(DEV)
        units.append(dict(y=y, scale=(g2 * uis).contiguous(), shift=(b2 - um * g2 * uis).contiguous(), mean=um.contiguous(), invstd=uis,
                          slabs=torch.zeros((nslabs, 2, c1 - c0), device=DEV), c0=c0, c1=c1))
    reqs = [ops.StatRequest(u['y'], u['scale'], u['shift'], u['slabs'], u['c0'], ops.ACT_SILU) for u in units]
    f1ws = torch.empty(ops.fused_1x1_ws_bytes(M, C, K) // 4 + 16, dtype=torch.float32, device=DEV)
    dx, dx_plain = to_dev_nhwc(dx0, dt, ld=C + 16, off=8), to_dev_nhwc(dx0, dt, ld=C + 16, off=8)
    gw, gw_plain = torch.zeros((K, C, 1, 1), device=DEV), torch.zeros((K, C, 1, 1), device=DEV)
    mu = mean.contiguous()
    ops.run([ops.rec_conv1x1_bwd_fused(dz, None, yd, scale, shift, mu, invstd, c1_d, c2_d, xd, wpd, dx, gw, None, f1ws, accumulate_dx=acc, stats=reqs),
             ops.rec_conv1x1_bwd_fused(dz, None, yd, scale, shift, mu, invstd, c1_d, c2_d, xd, wpd, dx_plain, gw_plain, None, f1ws, accumulate_dx=acc)])
    torch.cuda.synchronize()
    assert torch.equal(dx, dx_plain) and torch.equal(gw, gw_plain)
    for u in units:
        Kc = u['c1'] - u['c0']
        dg, db, c1, c2 = (torch.zeros(Kc, device=DEV) for _ in range(4))
        ops.run([ops.rec_bn_bwd_finalize_slabs(u['slabs'], M, u['mean'], u['invstd'], dg, db, c1, c2)])
        ws2 = torch.empty(ops.bn_bwd_ws_floats(M, Kc), dtype=torch.float32, device=DEV)
        dg2, db2 = torch.zeros(Kc, device=DEV), torch.zeros(Kc, device=DEV)
        ops.run([ops.rec_bn_act_bwd(dx[..., u['c0']:u['c1']], u['y'], u['scale'], u['shift'], u['mean'], u['invstd'], None, dg2, db2, ws2)])
        r1, r2 = ops.bn_bwd_coeffs(ws2, M, Kc)
        torch.cuda.synchronize()
        for got, ref, what in ((dg, dg2, 'dgamma'), (db, db2, 'dbeta'), (c1, r1, 'c1'), (c2, r2, 'c2')):
            assert_close(got.cpu(), ref.cpu(), 1e-4, what)
    assert dx._base[..., :8].float().eq(7.0).all()


def test_fused_sgd_matches_torch_sgd_step_by_step():
    """hd_yolo_amd.optim.SGD (one launch for all tensors) against torch.optim.SGD on the reference's recipe (train.py:208-233: three
    groups, Nesterov, weight decay on one group only, lr / momentum rewritten every step by the warm-up): parameters and momentum
    buffers after each of 4 steps, a parameter without a gradient, odd sizes (scalar tail path), and state_dict interchange."""
    from hd_yolo_amd.optim import SGD
    g = torch.Generator().manual_seed(7)
    shapes = [(64, 32, 3, 3), (39,), (128,), (5, 7), (1,), (256, 128, 1, 1), (4099,)]
    mk = lambda: [torch.nn.Parameter(torch.randn(s, generator=torch.Generator().manual_seed(i)).to(DEV)) for i, s in enumerate(shapes)]
    pa, pb = mk(), mk()

    def build(cls, ps):
        o = cls(ps[:2], lr=0.01, momentum=0.9, nesterov=True)
        o.add_param_group({'params': ps[2:5], 'weight_decay': 5e-4})
        o.add_param_group({'params': ps[5:]})
        return o
    oa, ob = build(SGD, pa), build(torch.optim.SGD, pb)
    for step in range(4):
        for j, (ga, gb) in enumerate(zip(oa.param_groups, ob.param_groups)):
            ga['lr'] = gb['lr'] = 0.01 * (step + 1) / 4 if j < 2 else 0.1 - 0.02 * step
            ga['momentum'] = gb['momentum'] = 0.8 + 0.03 * step
        for i, (a, b) in enumerate(zip(pa, pb)):
            if i == 3 and step < 2:
                a.grad = b.grad = None                      # joins later: its buffer starts at step 2
                continue
            gr = torch.randn(a.shape, generator=g).to(DEV)
            a.grad, b.grad = gr.clone(), gr.clone()
        oa.step(); ob.step()
        for i, (a, b) in enumerate(zip(pa, pb)):
            assert torch.allclose(a, b, rtol=2e-6, atol=1e-7), (step, i)
            sa, sb = oa.state[a].get('momentum_buffer'), ob.state[b].get('momentum_buffer')
            assert (sa is None) == (sb is None)
            if sa is not None:
                assert torch.allclose(sa, sb, rtol=2e-6, atol=1e-7), (step, i)
    # checkpoints are interchangeable
    oc = build(torch.optim.SGD, mk())
    oc.load_state_dict(oa.state_dict())
    od = build(SGD, mk())
    od.load_state_dict(ob.state_dict())
    assert torch.allclose(od.state[od.param_groups[0]['params'][0]]['momentum_buffer'], ob.state[pb[0]]['momentum_buffer'])
    with pytest.raises(TypeError):
        bad = torch.nn.Parameter(torch.zeros(4))
        bad.grad = torch.zeros(4)
        SGD([bad], lr=0.1).step()


def test_stem_weight_gradient_with_fused_batchnorm_backward_is_bit_identical():
    """hdy_conv_wgrad_stem_fused (dy computed from dz, y and the BatchNorm coefficients while the tile is staged) against the two-launch
    path it replaces: hdy_bn_act_bwd writing dy, then hdy_conv_wgrad(stem) reading it — the same arithmetic and the same bf16 rounding
    point, so the weight gradient must be identical."""
    N, H, W, K = 3, 64, 128, 32
    dt = torch.bfloat16
    img = rnd((N, 3, H, W), 1).to(DEV)
    prep = torch.zeros((N, H + 4, W + 4, 4), dtype=dt, device=DEV)
    ops.run([ops.rec_stem_prep(img, prep)])
    Ho, Wo = H // 2, W // 2
    y = rnd((N, Ho, Wo, K), 2).to(dt).to(DEV)
    dz = rnd((N, Ho, Wo, K), 3).to(dt).to(DEV)
    scale, shift = (rnd((K,), 4).abs() + 0.5).to(DEV), rnd((K,), 5).to(DEV)
    mean, invstd = rnd((K,), 6, 0.1).to(DEV), (rnd((K,), 7).abs() + 0.5).to(DEV)
    M = N * Ho * Wo
    ws_bn = torch.empty(ops.bn_bwd_ws_floats(M, K), dtype=torch.float32, device=DEV)
    dy = torch.empty_like(y)
    dg, db = torch.zeros(K, device=DEV), torch.zeros(K, device=DEV)
    ops.run([ops.rec_bn_act_bwd(dz, y, scale, shift, mean, invstd, dy, dg, db, ws_bn)])
    c1, c2 = ops.bn_bwd_coeffs(ws_bn, M, K)
    ws = torch.empty(ops.wgrad_ws_bytes(N, H, W, 3, K, 6, 6, 2, 2, dt, stem=True) // 4 + 16, dtype=torch.float32, device=DEV)
    g_ref = torch.zeros((K, 3, 6, 6), dtype=torch.float32, device=DEV)
    ops.run([ops.rec_conv_wgrad(prep, dy, g_ref, None, 6, 6, 2, 2, ws, stem_hw=(H, W))])
    assert ops.wgrad_stem_fused_ok(N, H, W, K, dt)
    g_fused = torch.full((K, 3, 6, 6), 7.0, dtype=torch.float32, device=DEV)
    ops.run([ops.rec_conv_wgrad_stem_fused(prep, dz, y, scale, shift, mean, invstd, c1, c2, (H, W), g_fused, None, ws)])
    assert torch.equal(g_fused, g_ref) and g_ref.abs().max().item() > 0
    assert not ops.wgrad_stem_fused_ok(N, H, W, 48, dt) and not ops.wgrad_stem_fused_ok(N, 60, 60, K, dt)



def test_det_targets_equals_the_tensor_expressions():
    """hdy_det_targets against the eager form it replaces (yolo_head.py:217-222: xyxy -> xywh rows and the one-hot class table without
    its column 0), labels outside 1..nc included."""
    from metayolo.models.utils_torch import one_hot_labels
    g = torch.Generator().manual_seed(3)
    nt, nc = 1000, 8
    xy = torch.rand((nt, 2), generator=g) * 0.8
    boxes = torch.cat([xy, xy + torch.rand((nt, 2), generator=g) * 0.2], 1).to(DEV)
    img = torch.randint(0, 64, (nt,), generator=g).float().to(DEV)
    labels = torch.randint(-1, nc + 3, (nt,), generator=g).to(DEV)
    gts, tcls = ops.det_targets(boxes, img, labels, nc)
    want = torch.stack([img, (boxes[:, 0] + boxes[:, 2]) / 2, (boxes[:, 1] + boxes[:, 3]) / 2, boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]], 1)
    assert torch.equal(gts, want)
    assert torch.equal(tcls, one_hot_labels(labels, nc)[:, 1:].float())
    e = ops.det_targets(boxes[:0], img[:0], labels[:0], nc)
    assert e[0].shape == (0, 5) and e[1].shape == (0, nc)


AB_CASES = [
    # (switch that disables the specialised kernel, its dispatch name, what, N, H, W, C, K, stride): shapes with >= 2x the kernel's grid in tiles
    ('HDY_NO_CONV3X3', 'conv3x3_c64', 'fwd', 8, 128, 128, 64, 64, 1),              # 1024 tiles of 8x16 on 512 workgroups
    ('HDY_NO_CONV3X3', 'conv3x3_c32', 'fwd', 6, 128, 128, 32, 32, 1),              # 768 tiles, three workgroups per CU
    ('HDY_NO_CONV3X3S2', 'conv3x3s2_c32', 'fwd', 6, 256, 256, 32, 64, 2),          # 1536 tiles of 4x16 outputs on 768 workgroups
    ('HDY_NO_DGRAD_S2', 'dgrad3x3s2_k64c32', 'dgrad', 16, 256, 256, 32, 64, 2),    # 1024 dy tiles of 8x16 on 512 workgroups
    ('HDY_NO_CONV3X3', 'conv3x3_c64', 'dgrad', 8, 128, 128, 64, 64, 1),
    ('HDY_NO_DGRAD_S2', 'dgrad3x3s2_k128c64', 'dgrad', 6, 256, 256, 64, 128, 2),   # 1536 dy tiles of 4x16 on 512 workgroups
    ('HDY_NO_CONV3X3S2', 'conv3x3s2_c64', 'fwd', 6, 256, 256, 64, 128, 2),         # 1536 tiles of 4x16 outputs on 256 eight-wave workgroups
    ('HDY_NO_CONV3X3_C128', 'conv3x3_c128', 'fwd', 5, 128, 128, 128, 128, 1),      # 1280 tiles of 8x8 on 256 streams x 2 channel halves (round 6)
    ('HDY_NO_CONV3X3_C128', 'conv3x3_c128', 'dgrad', 5, 128, 128, 128, 128, 1),
]


@pytest.mark.parametrize('case', AB_CASES, ids=[f'{c[1]}-{c[2]}' for c in AB_CASES])
def test_specialised_kernel_agrees_with_generic_beyond_one_grid(case):
    """Every patch- / filter-resident kernel against the generic implicit GEMM on the SAME inputs (its switch flips the dispatch), at more than
    twice its persistent grid in tiles, forward WITH BatchNorm sums and with the scale / shift / SiLU epilogue: outputs within one bf16 ulp
    of each other, slab sums within 1e-3, the fast kernel also against torch fp32 on the CPU; the dispatch log must show which kernel ran."""
    sw, name, what, N, H, W, C, K, stride = case
    dt = torch.bfloat16
    R, pad = 3, 1
    Ho, Wo = ops.out_dim(H, R, stride, pad), ops.out_dim(W, R, stride, pad)
    w = rnd((K, C, R, R), 2, (3.0 / (C * R * R)) ** 0.5)
    wq, wdev = q(w, dt), w.to(DEV)
    sc, sh = (rnd((K,), 3).abs() + 0.5).to(DEV), rnd((K,), 4).to(DEV)
    if what == 'fwd':
        x = q(rnd((N, C, H, W), 1), dt)
        xd = to_dev_nhwc(x, dt)
        wp = ops.pack_alloc(K, C, R, R, stride, pad, ops.PACK_FWD, dt, DEV)
        ops.run([ops.rec_pack(wdev, None, stride, pad, ops.PACK_FWD, wp)])
        res = {}
        for off in (0, 1):
            with _lib.option(sw, off):
                mt = ops.stat_slabs(N, H, W, C, K, R, R, stride, pad, dt)
                stats = torch.full((mt, 2, K), float('nan'), dtype=torch.float32, device=DEV)
                y = torch.empty((N, Ho, Wo, K), dtype=dt, device=DEV)
                y2 = torch.empty((N, Ho, Wo, K), dtype=dt, device=DEV)
                _lib.dispatch_log(reset=True)
                ops.run([ops.rec_conv_fwd(xd, wp, y, K, R, R, stride, pad, stats=stats),
                         ops.rec_conv_fwd(xd, wp, y2, K, R, R, stride, pad, scale=sc, shift=sh, act=ops.ACT_SILU)])
                log = _lib.dispatch_log(reset=True)
                assert (name in log) == (off == 0) and len(log) == 2 and log[0] == log[1], (off, log)
                res[off] = (y.float(), y2.float(), stats.sum(0))
        ref = F.conv2d(x, wq, None, stride, pad)
        assert_close(from_dev_nhwc(res[0][0].to(dt)), ref, TOL[dt], f'{name} vs torch')
        assert_close(from_dev_nhwc(res[0][1].to(dt)), F.silu(ref * sc.cpu().view(1, -1, 1, 1) + sh.cpu().view(1, -1, 1, 1)), TOL[dt] * 2, f'{name} epilogue vs torch')
        for i, what_ in ((0, 'raw output'), (1, 'SiLU epilogue')):
            a, b = res[0][i], res[1][i]
            ulp = (a - b).abs() / (b.abs().clamp_min(2.0 ** -10) * 2.0 ** -7)            # bf16: 8 significant bits
            assert ulp.max().item() <= 1.01, f'{name} {what_}: {ulp.max().item():.2f} bf16 ulps from the generic kernel'
        assert_close(res[0][2].cpu(), res[1][2].cpu(), 1e-3, f'{name} BatchNorm sums vs generic')
        assert_close(res[0][2][0].cpu(), ref.sum((0, 2, 3)), 1e-3, f'{name} BatchNorm sum vs torch')
    else:
        dy = q(rnd((N, K, Ho, Wo), 5), dt)
        dyd = to_dev_nhwc(dy, dt)
        wpd = ops.pack_alloc(K, C, R, R, stride, pad, ops.PACK_DGRAD, dt, DEV)
        ops.run([ops.rec_pack(wdev, None, stride, pad, ops.PACK_DGRAD, wpd)])
        res = {}
        for off in (0, 1):
            with _lib.option(sw, off):
                dx = torch.empty((N, H, W, C), dtype=dt, device=DEV)
                _lib.dispatch_log(reset=True)
                ops.run([ops.rec_conv_dgrad(dyd, wpd, dx, R, R, stride, pad)])
                log = _lib.dispatch_log(reset=True)
                assert (name in log) == (off == 0), (off, log)
                res[off] = dx.float()
        ref = F.conv_transpose2d(dy, wq, None, stride, pad, output_padding=(H - ((Ho - 1) * stride - 2 * pad + R), W - ((Wo - 1) * stride - 2 * pad + R)))
        assert_close(from_dev_nhwc(res[0].to(dt)), ref, TOL[dt], f'{name} vs torch')
        ulp = (res[0] - res[1]).abs() / (res[1].abs().clamp_min(2.0 ** -10) * 2.0 ** -7)
        assert ulp.max().item() <= 1.01, f'{name}: {ulp.max().item():.2f} bf16 ulps from the generic kernel'


WGRAD_DEEP_CASES = [
    # N, H, W, C, K, R, stride, pad — shapes of the deep-pipelined weight gradient (C % 64 == 0, K % 64 == 0, K >= 192, >= 8192 output pixels)
    (4, 96, 96, 128, 256, 3, 2, 1),      # 3x3 / stride 2: Q = 1152 = 4.5 column tiles, taps that leave the image
    (4, 96, 96, 64, 256, 3, 2, 1),       # C = 64: one tap per 64-column sub-tile, Q = 576
    (12, 53, 53, 64, 384, 3, 2, 1),      # K = 1.5 row tiles; 27 x 27 output pixels per image (8748 in all): the last stage of the last split is ragged
    (2, 72, 72, 64, 256, 5, 1, 2),       # 25 taps (Q = 1600), stride 1 (3x3 / stride 1 belongs to the patch-resident kernel)
    (4, 96, 96, 192, 192, 3, 2, 1),      # round 6: K = 192 = three quarters of one 256-row tile (yolov5m's neck downsample), three channel blocks per tap
    (4, 96, 96, 64, 320, 3, 2, 1),       # K = 1.25 row tiles: the second one a quarter full
]


@pytest.mark.parametrize('case', WGRAD_DEEP_CASES)
def test_deep_pipelined_weight_gradient(case):
    """conv_wgrad_deep.hip (256 x 256 output tile, pixels streamed through conv_deep's pipeline, both operands as transposed LDS reads):
    stacked gradients (grad_a / grad_b), accumulate, pitched x and dy, against torch autograd on the CPU — and against the generic kernel on
    the same operands (same products, different summation order: 1e-3 of the gradient's scale)."""
    N, H, W, C, K, R, stride, pad = case
    dt = torch.bfloat16
    _, _, log_w = conv_case(case, dt)
    assert 'wgrad_deep' in log_w, log_w
    x = q(rnd((N, C, H, W), 21), dt)
    Ho, Wo = ops.out_dim(H, R, stride, pad), ops.out_dim(W, R, stride, pad)
    dy = q(rnd((N, K, Ho, Wo), 22), dt)
    xd, dyd = to_dev_nhwc(x, dt), to_dev_nhwc(dy, dt)
    res = []
    for off in (0, 1):
        with _lib.option('HDY_NO_WGRAD_DEEP', off):
            ws = torch.empty(ops.wgrad_ws_bytes(N, H, W, C, K, R, R, stride, pad, dt) // 4 + 1, dtype=torch.float32, device=DEV)
            g = torch.zeros((K, C, R, R), dtype=torch.float32, device=DEV)
            _lib.dispatch_log(reset=True)
            ops.run([ops.rec_conv_wgrad(xd, dyd, g, None, R, R, stride, pad, ws)])
            assert ('wgrad_deep' in _lib.dispatch_log(reset=True)) == (off == 0)
            res.append(g.cpu())
    assert_close(res[0], res[1], 1e-3, 'deep vs generic weight gradient')


# ---------------------------------------------------------------------------------------------- compiled launch lists (csrc/exec.hip)
def test_compiled_launch_list_equals_the_python_loop():
    """ops.Program (hdy_exec_run: the whole list issued from C, forks onto the side stream and joins by the library's events) against ops.run
    (the Python loop) on a list with a fork whose result the main stream needs behind the join and a host callback in the middle (two
    segments); a failing item raises with the entry point's own message.  (float arguments — bit-pattern words — travel through the model tests:
    hdy_bn_finalize's eps / momentum against the reference goldens.)"""
    dt = torch.bfloat16
    M, K = 4096, 64

    def build(side):
        g = torch.Generator().manual_seed(5)
        a = torch.randn((1, 1, M, K), generator=g).to(DEV, dt)
        b = torch.randn((1, 1, M, K), generator=g).to(DEV, dt)
        c = torch.randn((1, 1, M, K), generator=g).to(DEV, dt)
        src = torch.arange(K, dtype=torch.float32, device=DEV)
        dst = torch.zeros(K, device=DEV)
        half = torch.full((1,), 0.5, device=DEV)
        seen = []
        recs = [ops.rec_add_inplace(a, b),                                               # a += b
                ('@fork', side, [ops.rec_add_inplace(c, a), ops.rec_add_inplace(c, a)], 3),   # side: c += 2a (needs the a of the line above)
                ('hdy_scale_inplace', (b.data_ptr(), b.numel(), half.data_ptr(), ops.dcode(dt)), (b, half)),      # main meanwhile: b *= 0.5
                ('@call', lambda: seen.append(1)),
                ops.rec_copy_f32(src, dst),
                ('@join', side, 3),
                ops.rec_add_inplace(b, c)]                                               # b += c: needs the fork's result
        return recs, (a, b, c, dst), seen

    results = []
    for compiled in (False, True):
        side = ops.SideStream(DEV)
        recs, tensors, seen = build(side)
        for _ in range(3):                       # replayed like a plan's list (events re-recorded every time)
            if compiled:
                prog = ops.Program(recs)
                assert [s[0] for s in prog.segments] == ['words', 'call', 'words']
                prog.run()
            else:
                ops.run(recs)
        torch.cuda.synchronize()
        assert len(seen) == 3
        results.append([t.float().cpu() for t in tensors])
    for x, y in zip(*results):
        assert torch.equal(x, y)
    assert results[0][3].tolist() == list(range(K))
    bad = ops.Program([ops.rec_add_inplace(tensors[0], tensors[1])])
    bad.segments[0][1][4] = 0                                                            # null second operand
    with pytest.raises(_lib.HdyError, match='add_inplace'):
        bad.run()
    with pytest.raises(_lib.HdyError, match='cannot be listed'):
        ops.Program([('hdy_version', ())])
