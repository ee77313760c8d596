"""hd_yolo_amd — MI355X-native (gfx950) implementation of hd_yolo's metayolo detection hot path.

Layout: csrc/ (HIP kernels + C ABI, include/hdyolo.h), _lib.py / ops.py (ctypes binding, launch records), plan.py /
engine.py (static forward/backward launch lists, gradient store, autograd hook-in), parallel.py (RCCL data parallel),
metayolo/ (the reference's module surface).  There is no CPU execution path: the CPU oracle lives in /oracle (tests only).
"""
import os

# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  RCCL brings its own streams: with the default the weight-gradient
# side stream then shares a queue with the main stream and the two launch lists run one after the other (measured with one rank over
# RCCL: 14.69 ms per train step against 13.82 with 8 queues).  Must be in the environment before the HIP runtime starts.
_queues_preset = os.environ.get('GPU_MAX_HW_QUEUES')
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')


def _hip_runtime_already_up():
    import sys
    t = sys.modules.get('torch')
    try:
        return bool(t is not None and t.cuda.is_initialized())
    except Exception:          # noqa: BLE001 — a torch without the cuda module, half-imported torch: nothing has started the runtime
        return False


# False: the runtime had started before this import with fewer than 8 queues in its environment (bench.py reports it on every line)
HW_QUEUES_IN_TIME = not (_hip_runtime_already_up() and (_queues_preset is None or int(_queues_preset or 0) < 8))
if not HW_QUEUES_IN_TIME:
    # the setdefault above is a no-op for a runtime that has already read its environment: say so LOUDLY, the symptom (the two launch lists of a
    # training step running one after the other as soon as RCCL brings its streams, +6 % per step) is silent
    import warnings
    warnings.warn('hd_yolo_amd: the HIP runtime was initialised before this package was imported and GPU_MAX_HW_QUEUES was '
                  f'{"unset (HIP default: 4)" if _queues_preset is None else _queues_preset} at that time.  With fewer than 8 hardware queues the '
                  'weight-gradient stream shares a queue with the main stream once RCCL creates its own (multi-GPU runs): export '
                  'GPU_MAX_HW_QUEUES=8 before the process touches the GPU, or import hd_yolo_amd before torch.cuda is initialised.',
                  RuntimeWarning, stacklevel=2)
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
