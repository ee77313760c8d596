"""N = 2 data-parallel path on one MI355X: two processes (gloo backend, both on cuda:0 — RCCL needs one device per rank,
which a 1-GPU box cannot give) run HIP-plan training steps through hd_yolo_amd.parallel.DataParallel.
Checks: parameters start identical (rank-0 broadcast), the applied gradient is the SUM of the per-rank gradients
(SURVEY.md §8e), and parameters stay bit-identical across ranks after optimizer steps."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), YOLOv5_VERBOSE='false',
                      HDY_GRAD_BUCKET_MB='1')              # yolov5n has 7 MB of gradients: several overlapped buckets per backward
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from hd_yolo_amd import synth
    from hd_yolo_amd.parallel import DataParallel
    from metayolo.models.yolo import Model
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        dev = torch.device('cuda', 0)
        torch.manual_seed(10 + rank)
        model = Model(synth.make_cfg('n', 2), synth.make_hyp())
        sd = synth.synth_state_dict(synth.shapes_of(model), seed=rank)          # ranks differ before the broadcast
        model.load_state_dict(sd, strict=False)
        model = model.to(dev).train()
        net = DataParallel(model, nbuckets=3)
        net.reducer.MIN_BYTES = 1 << 20
        start = torch.cat([p.detach().flatten() for p in model.parameters()]).cpu()
        x = synth.synth_images(2, 64, seed=20 + rank).to(dev)
        tg = synth.synth_targets(2, 64, 2, nmin=3, nmax=8, seed=30 + rank)
        # local gradient without the hook, for the sum check
        eng = model._eng()
        hooks, eng.grad_hooks = eng.grad_hooks, []
        bhooks, eng.bucket_hooks = eng.bucket_hooks, []
        losses, _ = net(x, tg)
        losses['det']['det_loss'].backward()
        local = torch.cat([p.grad.flatten() for p in model.parameters()]).cpu().clone()
        for p in model.parameters():
            p.grad = None
        for m in model.modules():                     # undo the BN statistics update of the probe step
            if isinstance(m, torch.nn.BatchNorm2d):
                m.load_state_dict({k: v for k, v in sd_bn(m).items()})
        eng.grad_hooks, eng.bucket_hooks = hooks, bhooks
        opt = torch.optim.SGD(model.parameters(), lr=1e-3, momentum=0.9)
        losses, _ = net(x, synth.synth_targets(2, 64, 2, nmin=3, nmax=8, seed=30 + rank))
        losses['det']['det_loss'].backward()
        reduced = torch.cat([p.grad.flatten() for p in model.parameters()]).cpu().clone()
        plan = next(iter(eng.plans.values()))
        assert len(plan.grad_marks) >= 4 and net.reducer.calls >= 4, (len(plan.grad_marks), net.reducer.calls)     # issued during backward
        covered = sorted((a, b) for _, a, b in plan.grad_marks)
        assert covered[0][0] == 0 and covered[-1][1] == eng.store.numel and all(x[1] == y[0] for x, y in zip(covered, covered[1:]))
        opt.step()
        opt.zero_grad(set_to_none=True)
        losses, _ = net(x, synth.synth_targets(2, 64, 2, nmin=3, nmax=8, seed=30 + rank))
        losses['det']['det_loss'].backward()
        opt.step()
        end = torch.cat([p.detach().flatten() for p in model.parameters()]).cpu()
        q.put((rank, start.numpy(), local.numpy(), reduced.numpy(), end.numpy()))       # by value: the worker exits right after
    finally:
        dist.destroy_process_group()


def sd_bn(m):
    return {'weight': m.weight.detach(), 'bias': m.bias.detach(), 'running_mean': m.running_mean, 'running_var': m.running_var,
            'num_batches_tracked': m.num_batches_tracked}


@pytest.mark.timeout(300)
def test_two_ranks_share_gradients_and_stay_in_sync():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, s0, l0, r0, e0), (_, s1, l1, r1, e1) = [(r, *map(torch.from_numpy, arrs)) for r, *arrs in out]
    assert torch.equal(s0, s1), 'rank 0 state was not broadcast'
    assert torch.equal(r0, r1), 'ranks disagree on the reduced gradient'
    torch.testing.assert_close(r0, l0 + l1, rtol=1e-4, atol=1e-6)       # SUM of the per-rank gradients, no division
    assert torch.equal(e0, e1), 'parameters drifted apart'
    assert not torch.equal(e0, s0)


def _syncbn_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), YOLOv5_VERBOSE='false')
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from hd_yolo_amd import synth
    from hd_yolo_amd.parallel import DataParallel
    from metayolo.models.yolo import Model
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        dev = torch.device('cuda', 0)
        model = Model(synth.make_cfg('n', 2), synth.make_hyp())
        model.load_state_dict(synth.synth_state_dict(synth.shapes_of(model), seed=0), strict=False)
        model = model.to(dev).train()
        net = DataParallel(model, sync_bn=True)
        x = synth.synth_images(2, 64, seed=20 + rank).to(dev)
        _, dets = model._eng().forward(x, True, torch.float32)
        g = torch.Generator().manual_seed(77)
        ws = [torch.randn((4,) + tuple(d.shape[1:]), generator=g) for d in dets]           # one weight tensor per level for the FULL batch of 4
        loss = sum((d * w[2 * rank:2 * rank + 2].to(dev)).sum() for d, w in zip(dets, ws))
        loss.backward()
        grads = {k: p.grad.detach().cpu().numpy().copy() for k, p in model.named_parameters()}             # numpy: pickled by value
        stats = {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items() if 'running_' in k}
        q.put((rank, [d.detach().cpu().numpy().copy() for d in dets], grads, stats))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_sync_batchnorm_two_ranks_equal_one_process_with_the_whole_batch():
    """train.py --sync-bn (reference: train.py:281-283): with SyncBatchNorm two ranks of two tiles each must produce what ONE process
    produces on the four tiles — logits per tile, BatchNorm running statistics, and (for a loss that is a plain sum over tiles, so that
    the data-parallel SUM of gradients is the full-batch gradient) every parameter gradient."""
    from hd_yolo_amd import synth
    from metayolo.models.yolo import Model
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_syncbn_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        r, dets, grads, stats = q.get(timeout=300)
        got[r] = ([torch.from_numpy(d) for d in dets], {k: torch.from_numpy(v) for k, v in grads.items()}, {k: torch.from_numpy(v) for k, v in stats.items()})
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    dev = torch.device('cuda', 0)
    model = Model(synth.make_cfg('n', 2), synth.make_hyp())
    model.load_state_dict(synth.synth_state_dict(synth.shapes_of(model), seed=0), strict=False)
    model = model.to(dev).train()
    x = torch.cat([synth.synth_images(2, 64, seed=20 + r) for r in range(2)]).to(dev)
    _, dets = model._eng().forward(x, True, torch.float32)
    g = torch.Generator().manual_seed(77)
    ws = [torch.randn((4,) + tuple(d.shape[1:]), generator=g) for d in dets]
    sum((d * w.to(dev)).sum() for d, w in zip(dets, ws)).backward()

    def rel(a, b):
        return ((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-12)).item()
    for r in range(2):
        for l, d in enumerate(dets):
            assert rel(got[r][0][l], d[2 * r:2 * r + 2].detach().cpu()) < 1e-4, (r, l)
    sd = model.state_dict()
    for k, v in got[0][2].items():
        assert rel(v, sd[k].cpu()) < 1e-5, k
        assert torch.equal(v, got[1][2][k]), k                       # identical on both ranks
    worst = 0.0
    for k, p in model.named_parameters():
        assert torch.equal(got[0][1][k], got[1][1][k]), k            # the all-reduced gradient is the same everywhere
        worst = max(worst, rel(got[0][1][k], p.grad.cpu()))
    assert worst < 2e-3, worst
    # and without SyncBatchNorm the per-rank statistics differ from the whole-batch ones (the test would not notice a no-op otherwise)
    m2 = Model(synth.make_cfg('n', 2), synth.make_hyp())
    m2.load_state_dict(synth.synth_state_dict(synth.shapes_of(m2), seed=0), strict=False)
    m2 = m2.to(dev).train()
    _, d2 = m2._eng().forward(x[:2].contiguous(), True, torch.float32)
    assert rel(d2[0].detach().cpu(), dets[0][:2].detach().cpu()) > 1e-3


def _bench_size_grads(reducer=None, steps=1):
    """flat gradient of one yolov5s / 8-class / batch 64 / 640x640 bf16 training step (BASELINE configs[1]) with an optional gradient hook"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.setdefault('YOLOv5_VERBOSE', 'false')
    from hd_yolo_amd import synth
    from metayolo.models.yolo import Model
    dev = torch.device('cuda', 0)
    model = Model(synth.make_cfg('s', 8), synth.make_hyp())
    model.load_state_dict(synth.synth_state_dict(synth.shapes_of(model), seed=0), strict=False)
    model = model.to(dev).train()
    model.half()
    eng = model._eng()
    if reducer is not None:
        eng.grad_hooks.append(reducer)
        eng.bucket_hooks.append(reducer.bucket)
    x = synth.synth_images(64, 640, seed=0).to(dev)
    t = synth.synth_targets(64, 640, 8, seed=1)
    out = None
    for _ in range(steps):
        for p in model.parameters():
            p.grad = None
        losses, _ = model(x, t)
        losses['det']['det_loss'].backward()
        torch.cuda.synchronize()
        out = torch.cat([p.grad.flatten() for p in model.parameters()]).clone()
    plan = next(iter(eng.plans.values()))
    return out, plan


@pytest.mark.timeout(600)
def test_overlapped_reduction_waits_for_every_writer_at_bench_size():
    """The N > 1 overlap on ONE GPU, falsifiable: the collective is replaced by a kernel on the communication stream that doubles the
    range (= SUM over two identical ranks).  With the reducer's stream dependencies in place every parameter gradient must be exactly
    twice the single-process gradient — a range doubled before its last writer (main stream or weight-gradient side stream) has finished
    ends up un-doubled or half-doubled.  Control: a reducer that takes a range before its writers are LAUNCHED (the whole buffer at the first
    mark) must NOT give twice the gradient.  (Round 3's control only dropped the stream waits; whether the doubling kernels then overtake the
    writers depends on which hardware queue HIP gives the communication stream — with more streams alive in the process they share the main
    stream's queue and run in order: the control passed or failed with the test order.)"""
    from hd_yolo_amd.parallel import GradAllReduce
    g, plan = _bench_size_grads()
    assert len(plan.grad_marks) >= 4                     # the 29 MB of yolov5s gradients leave the backward list in >= 4 ranges
    red = GradAllReduce(overlap=True, reduce_fn=lambda t: t.mul_(2.0))
    g2, _ = _bench_size_grads(red, steps=2)              # second step: buffers recycled, streams warm
    assert red.calls >= 8, red.calls                     # >= 4 per backward pass, issued during it
    assert torch.equal(g2, 2 * g), f'{(g2 - 2 * g).abs().max().item()} max deviation, {(g2 != 2 * g).float().mean().item():.3f} of the elements'
    # control: the first mark takes the WHOLE buffer — every later range is doubled before the launches that write it have been issued
    class Early(GradAllReduce):
        def bucket(self, store, a, b, side_stream=None):
            if not self.sent:
                self._side = side_stream
                self._send(store.cur, 0, store.cur.numel(), waits=self._streams(store.cur, side_stream))

    broken = Early(overlap=True, reduce_fn=lambda t: t.mul_(2.0))
    g3, _ = _bench_size_grads(broken, steps=2)
    assert broken.calls >= 2 and not torch.equal(g3, 2 * g), 'the check cannot see a range reduced before its writers ran'
    assert (g3 != 2 * g).float().mean().item() > 0.5          # everything but the first range


def _hnet_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), YOLOv5_VERBOSE='false',
                      HDY_GRAD_BUCKET_MB='1')
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.nn.functional as F
    from hd_yolo_amd import synth
    from hd_yolo_amd.hnet import HNet
    from hd_yolo_amd.parallel import DataParallel
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        dev = torch.device('cuda', 0)
        B, S, nc, ncls = 2, 64, 2, 3
        cfg = {'backbone': {'type': 'yolov5', 'cfg': synth.make_cfg('n', nc), 'hyp': synth.make_hyp()},
               'headers': {'seg': {'type': 'PanopticSeg', 'configs': {'num_classes': ncls, 'feature_maps': None, 'in_channels': None, 'scale_factor': 8,
                                                                       'resize_mode': 'bilinear', 'class_weight': None, 'roi_size': None}}}}
        m = HNet(cfg)
        m.detector.load_state_dict(synth.synth_state_dict(synth.shapes_of(m.detector), seed=rank), strict=False)      # ranks differ before the broadcast
        g = torch.Generator().manual_seed(5 + rank)
        with torch.no_grad():
            for k, p in m.headers.named_parameters():
                p.copy_(torch.randn(p.shape, generator=g) * (0.05 if p.dim() == 4 else 0.3) + (1.0 if p.dim() == 1 and k.endswith('weight') else 0.0))
        m = m.to(dev).train()
        net = DataParallel(m, nbuckets=3)
        net.reducer.MIN_BYTES = 1 << 20

        def batch():
            x = synth.synth_images(B, S, seed=20 + rank).to(dev)
            det_t = synth.synth_targets(B, S, nc, nmin=3, nmax=8, seed=30 + rank)
            lab = torch.randint(0, ncls, (B, S, S), generator=torch.Generator().manual_seed(40 + rank))
            masks = F.one_hot(lab, ncls).permute(0, 3, 1, 2).float().contiguous()
            targets = []
            for i, t in enumerate(det_t):
                anns = dict(t['anns'])
                anns['seg'] = [{'roi': torch.tensor([0.0, 0.0, S, S]), 'masks': masks[i].to(dev)}]
                targets.append({**t, 'anns': anns})
            return x, targets

        def step():
            for p in m.parameters():
                p.grad = None
            x, targets = batch()
            losses, _ = net(x, targets)
            (losses['det_det_loss'] + 2.0 * losses['seg_soft_iou_loss']).backward()
            return {k: p.grad.detach().cpu().numpy().copy() for k, p in m.named_parameters()}

        start = {k: p.detach().cpu().numpy().copy() for k, p in m.named_parameters()}
        eng = m._eng()
        hooks, eng.grad_hooks = eng.grad_hooks, []
        bhooks, eng.bucket_hooks = eng.bucket_hooks, []
        bn = {k: v.clone() for k, v in m.state_dict().items() if 'running_' in k or 'num_batches' in k}
        local = step()
        m.load_state_dict(bn, strict=False)                  # undo the BatchNorm statistics update of the probe step
        eng.grad_hooks, eng.bucket_hooks = hooks, bhooks
        reduced = step()
        q.put((rank, start, local, reduced))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_hnet_two_ranks_sum_detector_and_segmentation_header_gradients():
    """BASELINE configs[4] shape of the data-parallel path: HNet (detector plan + PanopticSeg header on its pyramid taps) under DataParallel,
    two ranks over gloo on one GPU.  The segmentation header's parameters live in the same flat gradient store as the detector's (their
    gradients are written by the header's own backward, outside the launch list): every parameter — backbone, neck, detection convs,
    connector ladders, class conv — must start from rank 0's value and receive the SUM of the two ranks' gradients."""
    import numpy as np
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_hnet_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=240) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, s0, l0, r0), (_, s1, l1, r1) = out
    seg_keys = [k for k in s0 if k.startswith('headers.')]
    assert len(seg_keys) >= 6 and any('detector.backbone' in k or k.startswith('detector.') for k in s0)
    for k in s0:
        assert np.array_equal(s0[k], s1[k]), f'{k}: rank 0 state was not broadcast'
        assert np.array_equal(r0[k], r1[k]), f'{k}: ranks disagree on the reduced gradient'
        want = l0[k] + l1[k]
        err = np.abs(r0[k] - want).max() / (np.abs(want).max() + 1e-12)
        assert err < 1e-4, f'{k}: reduced gradient is not the sum of the ranks ({err:.2e})'
    assert any(np.abs(l0[k]).max() > 0 for k in seg_keys)
