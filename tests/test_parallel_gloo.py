"""Data-parallel plumbing on CPU: two and eight processes over gloo (the N > 1 path of bench.py / train.py without a GPU).

Checks the properties the 8-GPU run relies on: rank 0's state is what every rank starts from; the flat-gradient
all-reduce is a SUM (no division: DDP's mean of the WORLD_SIZE-prescaled loss) and is identical on all
ranks whatever the bucket count; summed per-rank gradients of per-rank losses equal the single-process gradient of the
concatenated batch for a loss of the form sum-over-images (which DetLoss * bs is, per image row)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hd_yolo_amd import parallel


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


class _Store:
    def __init__(self, n):
        self.cur = torch.zeros(n)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)                       # ranks start different on purpose
        net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.BatchNorm2d(8), torch.nn.Conv2d(8, 4, 1))
        net[1].running_mean.add_(rank + 1.0)
        parallel.broadcast_state(net, 0)
        state = torch.cat([t.detach().reshape(-1).double() for t in list(net.parameters()) + list(net.buffers())])
        gathered = [torch.zeros_like(state) for _ in range(world)]
        dist.all_gather(gathered, state)
        same_start = all(torch.equal(gathered[0], g) for g in gathered)

        # per-rank shard of a global batch; loss = sum over images (like DetLoss * bs)
        g = torch.Generator().manual_seed(7)
        x_all = torch.randn((2 * world, 3, 8, 8), generator=g)
        net.eval()                                          # BN statistics are per rank in training; keep the check linear
        shard = x_all[rank * 2:(rank + 1) * 2]
        net.zero_grad()
        net(shard).pow(2).sum().backward()
        params = list(net.parameters())
        store = _Store(sum(p.numel() for p in params) + 5)   # deliberately not a multiple of the bucket alignment
        off = 0
        for p in params:
            store.cur[off:off + p.numel()] = p.grad.reshape(-1)
            off += p.numel()
        results = []
        for nb in (1, 3, 64):
            s2 = _Store(store.cur.numel())
            s2.cur.copy_(store.cur)
            parallel.GradAllReduce(nbuckets=nb)(s2)
            results.append(s2.cur.clone())
        # the overlapped path: ranges reported final from the end of the buffer while "backward" is still running, the rest at the end
        n = store.cur.numel()
        for min_bytes in (0, 1 << 20):
            s2 = _Store(n)
            s2.cur.copy_(store.cur)
            red = parallel.GradAllReduce()
            red.MIN_BYTES = min_bytes
            assert n > 200
            red.bucket(s2, n - 100, n)
            red.bucket(s2, 120, n - 100)
            red.bucket(s2, 40, 80)               # not adjacent to the previous range
            red(s2)                              # [0, 40) and [80, 120) were never marked
            assert red.sent == [] and red.pending is None
            results.append(s2.cur.clone())
        net.zero_grad()
        net(x_all).pow(2).sum().backward()
        ref = torch.cat([p.grad.reshape(-1) for p in params])
        # by value (numpy): a tensor on an mp.Queue is a shared-memory handle that dies with this process
        q.put((rank, same_start, [r[:ref.numel()].numpy().copy() for r in results], ref.numpy().copy(), results[0][ref.numel():].numpy().copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize('world', [2, 8])
def test_broadcast_and_sum_allreduce(world):
    """world 8 = the rank count of the SCALE run (one node of eight MI355X): same plumbing, gloo instead of RCCL, CPU tensors"""
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    out = [(r, s, [torch.from_numpy(a) for a in res], torch.from_numpy(ref), torch.from_numpy(tail)) for r, s, res, ref, tail in out]
    out.sort(key=lambda t: t[0])
    for rank, same_start, results, ref, tail in out:
        assert same_start, 'ranks did not start from rank 0 state'
        for r in results:
            torch.testing.assert_close(r, ref, rtol=1e-5, atol=1e-6)       # SUM of shards == full-batch gradient
            if world == 2:
                assert torch.equal(r, results[0])                          # a + b: bit-identical whatever the bucket count
            else:                                                          # eight addends: the ring's summation order moves with the chunking
                torch.testing.assert_close(r, results[0], rtol=1e-6, atol=1e-6)
        assert torch.equal(tail, torch.zeros_like(tail))
    for k in range(len(out[0][2])):
        assert all(torch.equal(out[0][2][k], o[2][k]) for o in out[1:]), 'ranks disagree after the all-reduce'


def test_bucket_bounds_cover_exactly():
    for n in (1, 1000, 1024, 7041205, 7041205 + 3):
        for nb in (1, 4, 7, 1000000):
            b = parallel.bucket_bounds(n, nb)
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(x[1] == y[0] for x, y in zip(b, b[1:]))
            assert all(lo < hi for lo, hi in b)
            assert len(b) <= max(nb, 1)


def test_expected_sends_is_what_the_reducer_issues():
    """GradAllReduce.expected_sends (what bench.py prints as `allreduce_expected`) against the calls the reducer really makes for the same marks:
    ranges under MIN_BYTES held back and merged with a contiguous neighbour, a range kept back that is NOT contiguous sent on its own, leftovers
    behind the list — on host tensors (no BUCKET_BYTES cut), with a recording stand-in for the collective."""
    n = 5_000_000
    mb = lambda x: int(x * (1 << 20)) // 4          # noqa: E731 — megabytes -> fp32 elements
    # completion order of a backward list: from the end of the buffer; one small range (merges with the next), one small non-contiguous range
    marks, hi = [], n
    for size in (mb(6.5), mb(1.0), mb(5.0), mb(0.5)):
        marks.append((hi - size, hi))
        hi -= size
    marks.append((mb(0.25), mb(0.75)))                # not contiguous with the range kept back: that one goes out alone
    store = _Store(n)
    got = []
    red = parallel.GradAllReduce(reduce_fn=lambda t: got.append((t.storage_offset(), t.storage_offset() + t.numel())))
    for a, b in marks:
        red.bucket(store, a, b)
    red(store)
    want = parallel.GradAllReduce.expected_sends(marks, n, split=False)
    assert got == want, (got, want)
    covered = sorted(got)
    assert covered[0][0] == 0 and covered[-1][1] == n and all(x[1] == y[0] for x, y in zip(covered, covered[1:]))      # tiles the buffer exactly once
    assert red.calls == len(want) and red.bytes == 4 * n
    # the GPU path cuts a call at BUCKET_BYTES: same coverage, no call above the cap
    cut = parallel.GradAllReduce.expected_sends([(0, 40_000_000)], 40_000_000)
    assert max(b - a for a, b in cut) * 4 <= parallel.GradAllReduce.BUCKET_BYTES and sum(b - a for a, b in cut) == 40_000_000


def test_late_hw_queue_setting_warns_loudly():
    """hd_yolo_amd/__init__.py: GPU_MAX_HW_QUEUES only counts if it is in the environment before the HIP runtime starts.  A caller that initialised
    torch.cuda first gets a RuntimeWarning instead of two silently serialised launch lists under RCCL."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ('import sys, warnings, torch\n'
            'torch.cuda.is_initialized = lambda: True\n'            # stands in for a caller that touched the GPU before the import
            'warnings.simplefilter("always")\n'
            'import hd_yolo_amd\n')
    env = {k: v for k, v in os.environ.items() if k != 'GPU_MAX_HW_QUEUES'}
    p = subprocess.run([sys.executable, '-c', code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and 'GPU_MAX_HW_QUEUES' in p.stderr and 'RuntimeWarning' in p.stderr, p.stderr[-2000:]
    p = subprocess.run([sys.executable, '-c', code], cwd=root, env=dict(env, GPU_MAX_HW_QUEUES='8'), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and 'GPU_MAX_HW_QUEUES' not in p.stderr, p.stderr[-2000:]
