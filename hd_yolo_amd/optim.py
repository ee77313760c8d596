"""torch.optim.SGD's update as ONE HIP launch over all parameter tensors (csrc/optim.hip).

Drop-in for the optimizer the reference builds (train.py:208-233: `SGD(g0, lr, momentum, nesterov=True)` + `add_param_group` for the
decayed weights and the biases; train.py:436-444 rewrites `lr` / `momentum` of every group during the warm-up; train.py:478 steps
it): same constructor arguments, `param_groups`, `state[p]['momentum_buffer']` and `state_dict()` layout, so checkpoints written with
either class load into the other.  Parameters and gradients must be contiguous fp32 CUDA tensors (the engine's gradients are views of
one flat fp32 buffer); anything else raises — there is no fallback path."""
import ctypes

import torch

from . import _lib


class SGD(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False):
        if nesterov and (momentum <= 0 or dampening != 0):
            raise ValueError('Nesterov momentum requires a momentum and zero dampening')
        super().__init__(params, dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay, nesterov=nesterov))
        self._key, self._table, self._ndesc, self._blocks = None, None, 0, 0

    def _build(self, rows, dev):
        lib = _lib.load()
        descs = (_lib.SgdDesc * len(rows))()
        blocks = 0
        for d, (p, g, buf, gi, first) in zip(descs, rows):
            d.p, d.g, d.buf = p.data_ptr(), g.data_ptr(), (buf.data_ptr() if buf is not None else None)
            d.n, d.group, d.first, d.first_block = p.numel(), gi, int(first), blocks
            blocks += lib.hdy_sgd_blocks(p.numel())
        host = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8)
        self._table, self._ndesc, self._blocks = host.to(dev), len(rows), blocks

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if len(self.param_groups) > 8:
            raise ValueError('hd_yolo_amd.optim.SGD: at most 8 parameter groups')
        nesterov = {bool(g['nesterov']) for g in self.param_groups}
        if len(nesterov) != 1:
            raise ValueError('hd_yolo_amd.optim.SGD: nesterov must be the same in every parameter group')
        rows, key, dev = [], [], None
        for gi, group in enumerate(self.param_groups):
            for p in group['params']:
                g = p.grad
                if g is None:
                    continue
                if not (p.is_cuda and g.is_cuda and p.dtype == torch.float32 and g.dtype == torch.float32 and p.is_contiguous() and g.is_contiguous()):
                    raise TypeError('hd_yolo_amd.optim.SGD needs contiguous fp32 CUDA parameters and gradients')
                dev = p.device
                st = self.state[p]
                buf, first = st.get('momentum_buffer'), False
                if group['momentum'] != 0 and buf is None:
                    buf = st['momentum_buffer'] = torch.empty_like(p, memory_format=torch.contiguous_format)
                    first = True
                if group['momentum'] == 0:
                    buf = None
                rows.append((p, g, buf, gi, first))
                key.append((p.data_ptr(), g.data_ptr(), 0 if buf is None else buf.data_ptr(), gi, first))
        if not rows:
            return loss
        if key != self._key:
            self._build(rows, dev)
            # a table with first-step rows is valid for this step only
            self._key = None if any(r[4] for r in rows) else key
        n = len(self.param_groups)
        arr = lambda name: (ctypes.c_float * n)(*[float(g[name]) for g in self.param_groups])
        lib = _lib.load()
        rc = lib.hdy_sgd_step(self._table.data_ptr(), self._ndesc, self._blocks, arr('lr'), arr('momentum'), arr('dampening'), arr('weight_decay'),
                              n, int(nesterov.pop()), torch.cuda.current_stream(dev).cuda_stream)
        if rc:
            raise _lib.HdyError(f'hdy_sgd_step failed (status {rc}): {lib.hdy_last_error().decode()}')
        # the kernel wrote the parameters through raw pointers: tell autograd / every `_version`-keyed cache (segrun.PackCache, ops.PackTable,
        # ops.BnEvalTable) that they changed, as an in-place torch update would have
        torch._C._increment_version([p for p, _, _, _, _ in rows] + [buf for _, _, buf, _, _ in rows if buf is not None])
        return loss
