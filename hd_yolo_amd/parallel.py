"""Data-parallel training over the 8 MI355X of a node: one process per GPU, RCCL (torch.distributed "nccl") over xGMI.

What the reference does (train.py:331 DistributedDataParallel(find_unused_parameters=True), :467 `loss *= WORLD_SIZE`,
25 MB autograd-hook buckets, buffer broadcast every forward) versus here:

  * gradients of all parameters already sit in ONE flat fp32 buffer (engine.GradStore.cur), laid out once; the
    all-reduce is one large contiguous RCCL call (more only above 64 MB) on a side stream, not ~180 per-tensor hooks.  xGMI is
    point-to-point (7 links x ~153 GB/s per GPU), so a few large messages that RCCL can spread over all links beat
    many small ring steps.
  * reduction op is SUM with no division: DDP averages gradients of the WORLD_SIZE-prescaled loss, which is exactly the
    plain sum of per-rank gradients; callers therefore do NOT pre-scale the loss.
  * BatchNorm statistics stay per rank and no buffers are broadcast per step (rank 0's are the ones checkpointed, as in
    the reference where only rank 0 saves: train.py:500,529).
  * initial state: rank 0's parameters and buffers are broadcast once, flattened into two messages.

Works with any torch.distributed backend; the CPU tests run it over gloo with world_size 2.
"""
import os

import torch
import torch.distributed as dist


def flat_views(tensors):
    return [t.detach().view(-1) for t in tensors]


def broadcast_state(module, src=0):
    """One-time broadcast of parameters and floating/integer buffers from `src` (flattened per dtype)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    groups = {}
    for t in list(module.parameters()) + list(module.buffers()):
        groups.setdefault(t.dtype, []).append(t)
    for dtype, ts in groups.items():
        flat = torch.cat([t.detach().reshape(-1) for t in ts])
        dist.broadcast(flat, src)
        off = 0
        with torch.no_grad():
            for t in ts:
                n = t.numel()
                t.copy_(flat[off:off + n].view(t.shape))
                off += n


def bucket_bounds(numel, nbuckets, align=1024):
    """Split [0, numel) into `nbuckets` nearly equal, `align`-element aligned ranges."""
    nbuckets = max(1, min(nbuckets, (numel + align - 1) // align))
    per = ((numel + nbuckets - 1) // nbuckets + align - 1) // align * align
    out, a = [], 0
    while a < numel:
        b = min(a + per, numel)
        out.append((a, b))
        a = b
    return out


class GradAllReduce:
    """Sum-all-reduce of the flat gradient buffer on a communication stream, overlapped with the backward pass.

    The backward launch list carries marks "flat range [a, b) is final" (plan.Plan._mark_buckets: ranges of a few MB, completing
    from the end of the buffer because the list runs the layers in reverse).  `bucket()` is called at such a mark: the
    communication stream waits for what the compute streams have been given so far (events, no host sync) and gets that range's
    all-reduce, while the main stream goes on with the earlier layers.  `__call__` runs after the list: it reduces whatever no
    mark covered and makes the main stream wait for the communication stream.  Adjacent ready ranges are sent as one call up to
    BUCKET_BYTES (xGMI is point-to-point: a few large messages beat many small ring steps)."""

    BUCKET_BYTES = 64 << 20       # upper bound of one RCCL call
    MIN_BYTES = 4 << 20           # a ready range smaller than this waits for its neighbour (fixed RCCL latency per call)

    def __init__(self, nbuckets=None, group=None, overlap=True, reduce_fn=None):
        """reduce_fn(range_tensor): replaces the collective (tests: a kernel on the communication stream that doubles the range stands in
        for the SUM over two identical ranks, so that a one-GPU box can check the stream ordering of the overlap)"""
        self.nbuckets, self.group, self.overlap = nbuckets, group, overlap
        self.reduce_fn = reduce_fn
        self.stream = None
        self.sent = []            # ranges already given to the communication stream in this backward pass
        self.pending = None       # a ready range not yet sent (too small on its own)
        self.calls = 0            # all_reduce calls issued (tests / diagnostics)
        self.bytes = 0            # bytes handed to them
        self.exposed_events = None   # a list: __call__ appends an event pair around the main stream's wait for the communication stream (bench.py)

    @classmethod
    def expected_sends(cls, marks, numel, split=True):
        """The all-reduce calls ONE backward pass issues for a plan whose marks (Plan.bucket_marks(): ranges in the order the backward
        list completes them) cover a flat buffer of `numel` fp32 elements: [(a, b)] in issue order — the hold-back of ranges under
        MIN_BYTES, the merge with a contiguous neighbour, the leftovers behind the list and the BUCKET_BYTES cut, exactly as bucket() /
        __call__ / _send do them on a GPU.  bench.py prints it next to the measured count, so that the first run with N > 1 ranks checks itself."""
        sent, pending = [], None
        for a, b in marks:
            p, pending = pending, None
            if p is not None and p[1] == a:
                a = p[0]
            elif p is not None and p[0] == b:
                b = p[1]
            elif p is not None:
                sent.append(p)
            if (b - a) * 4 >= cls.MIN_BYTES:
                sent.append((a, b))
            else:
                pending = (a, b)
        if pending is not None:
            sent.append(pending)
        x = 0
        for a, b in sorted(sent):
            if a > x:
                sent.append((x, a))
            x = max(x, b)
        if x < numel:
            sent.append((x, numel))
        if not split:                   # host tensors (gloo rehearsals on the CPU) go out uncut
            return sent
        return [(c, min(b, c + cls.BUCKET_BYTES // 4)) for a, b in sent for c in range(a, b, cls.BUCKET_BYTES // 4)]

    def _active(self):
        # HDY_FORCE_DIST=1: issue the collectives with one rank too (rehearsal of the N > 1 path over RCCL on a one-GPU box)
        if self.reduce_fn is not None:
            return True
        return dist.is_initialized() and (dist.get_world_size(self.group) > 1 or os.environ.get('HDY_FORCE_DIST') == '1')

    def _send(self, flat, a, b, waits):
        if flat.is_cuda:
            if self.stream is None:
                self.stream = torch.cuda.Stream(device=flat.device)
            for s in waits:
                if s is not None:
                    self.stream.wait_stream(s)
            with torch.cuda.stream(self.stream):
                for x in range(a, b, self.BUCKET_BYTES // 4):
                    self._reduce(flat[x:min(b, x + self.BUCKET_BYTES // 4)])
        else:
            self._reduce(flat[a:b])
        self.sent.append((a, b))

    def _reduce(self, t):
        if self.reduce_fn is not None:
            self.reduce_fn(t)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        self.calls += 1
        self.bytes += t.numel() * t.element_size()

    def bucket(self, store, a, b, side_stream=None):
        """Engine bucket hook: gradient elements [a, b) of store.cur are final once the work issued so far has run."""
        if not self._active() or not self.overlap:
            return
        flat = store.cur
        self._side = side_stream
        p, self.pending = self.pending, None
        if p is not None and p[1] == a:                 # contiguous with the range kept back: one message
            a = p[0]
        elif p is not None and p[0] == b:
            b = p[1]
        elif p is not None:
            self._send(flat, *p, waits=self._streams(flat, side_stream))
        if (b - a) * 4 >= self.MIN_BYTES:
            self._send(flat, a, b, waits=self._streams(flat, side_stream))
        else:
            self.pending = (a, b)

    def _streams(self, flat, side_stream):
        return [torch.cuda.current_stream(flat.device), side_stream] if flat.is_cuda else []

    def __call__(self, store):
        if not self._active():
            return
        flat = store.cur
        waits = self._streams(flat, getattr(self, '_side', None))
        if self.pending is not None:
            self._send(flat, *self.pending, waits=waits)
            self.pending = None
        # whatever no mark covered (overlap off, or a plan without marks): the gaps between the ranges already sent
        gaps, x = [], 0
        for a, b in sorted(self.sent):
            if a > x:
                gaps.append((x, a))
            x = max(x, b)
        if x < flat.numel():
            gaps.append((x, flat.numel()))
        if self.nbuckets and not self.sent:
            gaps = bucket_bounds(flat.numel(), self.nbuckets)
        for a, b in gaps:
            self._send(flat, a, b, waits=waits)
        self.sent = []
        if flat.is_cuda:
            main = torch.cuda.current_stream(flat.device)
            if self.exposed_events is not None:
                # elapsed time between the two = how long the main stream stood waiting for the collectives (nothing else lies between them)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(main)
                main.wait_stream(self.stream)
                e1.record(main)
                self.exposed_events.append((e0, e1))
            else:
                main.wait_stream(self.stream)


class DataParallel:
    """Thin wrapper with the call surface the entry points need (`model(x, targets)`, `.module`-style access through
    hdy_dp_module): broadcasts rank 0's state once and installs the flat all-reduce as the engine's gradient hook."""

    def __init__(self, model, nbuckets=None, overlap=True, sync_bn=False):
        self.hdy_dp_module = model
        broadcast_state(model, 0)
        if sync_bn:                        # train.py --sync-bn (reference: train.py:281-283): BatchNorm statistics over all ranks' tiles
            model._eng().sync_bn = True
        self.reducer = GradAllReduce(nbuckets, overlap=overlap)
        model._eng().grad_hooks.append(self.reducer)
        model._eng().bucket_hooks.append(self.reducer.bucket)

    def __call__(self, *a, **k):
        return self.hdy_dp_module(*a, **k)

    def __getattr__(self, name):
        return getattr(self.__dict__['hdy_dp_module'], name)
