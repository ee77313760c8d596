"""hd_yolo_amd — MI355X-native (gfx950) implementation of hd_yolo's metayolo detection hot path.

Layout: csrc/ (HIP kernels + C ABI, include/hdyolo.h), _lib.py / ops.py (ctypes binding, launch records), plan.py /
engine.py (static forward/backward launch lists, gradient store, autograd hook-in), parallel.py (RCCL data parallel),
metayolo/ (the reference's module surface).  There is no CPU execution path: the CPU oracle lives in /oracle (tests only).
"""
import os

# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  RCCL brings its own streams: with the default the weight-gradient
# side stream then shares a queue with the main stream and the two launch lists run one after the other (measured with one rank over
# RCCL: 14.69 ms per train step against 13.82 with 8 queues).  Must be in the environment before the HIP runtime starts.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: RCCL across processes needs it on this driver


def host_cpu_quota():
    """CPU cores this process may actually use (cgroup CFS quota), or os.cpu_count() when unlimited."""
    n = os.cpu_count() or 1
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:                      # cgroup v2: "<quota> <period>" or "max <period>"
            quota, period = f.read().split()
            if quota != 'max':
                n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        try:
            with open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us') as f, open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as g:
                q, p = int(f.read()), int(g.read())
                if q > 0:
                    n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return n


def limit_host_threads(max_threads=8):
    """Cap torch's intra-op CPU threads to the cgroup quota.  On a GPU box the visible core count (256) is far above the
    quota (16): the default OpenMP pool then spins on every tiny CPU op (target bookkeeping), burns the whole CFS quota and
    the launch thread gets throttled for tens of ms every 100 ms period — measured as a 38 ms stall every third training
    step.  The reference caps its threads for the same reason (metayolo/__init__.py:21,31)."""
    import torch
    ranks = max(1, int(os.environ.get('LOCAL_WORLD_SIZE', '1') or 1))       # one process per GPU shares the node's quota
    n = max(1, min(max_threads, host_cpu_quota() // ranks))
    if torch.get_num_threads() > n:
        torch.set_num_threads(n)
    os.environ.setdefault('OMP_NUM_THREADS', str(n))
    return n


limit_host_threads()
