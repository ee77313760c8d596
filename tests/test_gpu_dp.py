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
