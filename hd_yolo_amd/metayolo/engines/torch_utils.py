"""Host helpers the entry points use (reference: metayolo/engines/torch_utils.py:33-40 torch_distributed_zero_first,
:53-81 select_device, :84-88 time_sync, :151-173 to_device / collate_fn)."""
import os
import time
from contextlib import contextmanager

import torch
import torch.distributed as dist


@contextmanager
def torch_distributed_zero_first(local_rank: int):
    """Let rank 0 go first (dataset caching etc.), everyone else waits at a barrier."""
    if local_rank not in (-1, 0):
        dist.barrier()
    yield
    if local_rank == 0:
        dist.barrier()


def select_device(device='', batch_size=None):
    """'' or 'N' or 'N,M' -> torch.device('cuda:N').  There is no CPU path in this build."""
    if str(device).lower() == 'cpu':
        raise RuntimeError('hd_yolo_amd has no CPU execution path: pass a GPU index')
    if not torch.cuda.is_available():
        raise RuntimeError('no MI355X visible (torch.cuda.is_available() is False)')
    # The reference narrows CUDA_VISIBLE_DEVICES before its first CUDA query (torch_utils.py:60-62); here the runtime may already be
    # initialised (and a visibility variable set afterwards is ignored), so the first listed index is selected explicitly.
    idx = 0
    if str(device).strip():
        idx = int(str(device).replace(' ', '').split(',')[0])
        if idx < 0 or idx >= torch.cuda.device_count():
            raise RuntimeError(f'--device {device}: this process sees {torch.cuda.device_count()} GPU(s)')
    torch.cuda.set_device(idx)
    return torch.device('cuda', idx)


def time_sync():
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    return time.time()


def to_device(x, device):
    if isinstance(x, torch.Tensor):
        return x.to(device, non_blocking=True)
    if isinstance(x, dict):
        return {k: to_device(v, device) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(to_device(v, device) for v in x)
    return x


def collate_fn(batch):
    return tuple(zip(*batch))
