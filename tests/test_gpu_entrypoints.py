"""The two re-authored entry points run end to end on the MI355X path (small synthetic jobs)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(cmd, env=None, timeout=500):
    e = dict(os.environ, YOLOv5_VERBOSE='true', **(env or {}))
    p = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    return p.stdout + p.stderr


def test_train_py_trains_validates_checkpoints_and_resumes(tmp_path):
    base = [sys.executable, 'train.py', '--variant', 'n', '--nc', '2', '--batch-size', '8', '--imgsz', '128', '--steps-per-epoch', '6',
            '--val-batches', '1', '--project', str(tmp_path), '--name', 'run', '--exist-ok', '--log-every', '2']
    out = run(base + ['--epochs', '2'])
    assert 'epochs completed' in out and 'mAP@.5' in out
    ck = torch.load(tmp_path / 'run' / 'weights' / 'last.pt', map_location='cpu')
    assert ck['epoch'] == 2 and 'backbone.0.conv.weight' in ck['model'] and 'headers.det.m.2.bias' in ck['ema']
    assert all(torch.isfinite(v).all() for v in ck['model'].values() if v.dtype.is_floating_point)
    out = run(base + ['--epochs', '3', '--resume', '--weights', str(tmp_path / 'run' / 'weights' / 'last.pt')])
    assert 'epoch 2/2' in out and 'epoch 0/2' not in out
    out = run([sys.executable, 'val_nuclei.py', '--variant', 'n', '--nc', '2', '--imgsz', '128', '--batch-size', '4', '--batches', '2',
               '--weights', str(tmp_path / 'run' / 'weights' / 'last.pt')])
    assert 'fitness' in out


@pytest.mark.parametrize('extra', [[], ['--sync-bn']])
def test_train_py_two_ranks_gloo(tmp_path, extra):
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29531' if not extra else '29533', 'train.py', '--variant', 'n', '--nc', '2', '--batch-size', '8', '--imgsz', '64', '--epochs', '1',
           '--steps-per-epoch', '3', '--val-batches', '1', '--project', str(tmp_path), '--name', 'dp', '--exist-ok'] + extra
    out = run(cmd, env={'HDY_DIST_BACKEND': 'gloo'})
    assert 'epochs completed' in out


def test_train_py_multi_scale(tmp_path):
    """--multi-scale 0.5 (reference train.py:447-452, flag :610): the tile size changes from step to step (64..192 in multiples of 32)"""
    out = run([sys.executable, 'train.py', '--variant', 'n', '--nc', '2', '--batch-size', '4', '--imgsz', '128', '--epochs', '1', '--steps-per-epoch', '6',
               '--multi-scale', '0.5', '--noval', '--project', str(tmp_path), '--name', 'ms'])
    assert 'epochs completed' in out


def test_bench_py_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with no rendezvous in the environment (how the driver calls it) starts two rank processes itself;
    on a one-GPU box they share the card over gloo."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    p = subprocess.run([sys.executable, 'bench.py', '--gpus', '2', '--steps', '3', '--warmup', '1', '--batch', '4', '--variant', 's', '--size', '128',
                        '--no-roofline', '--no-cpu-baseline', '--no-infer'], cwd=ROOT, env=dict(env, YOLOv5_VERBOSE='false'), capture_output=True, text=True, timeout=500)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith('{')][-1])
    assert line['n_gpus'] == 2 and line['config']['world_size'] == 2 and line['config']['global_batch'] == 8
    assert line['config']['backend'] in ('gloo', 'nccl') and line['value'] > 0
    # yolov5s: 28 MB of gradients leave the backward list in >= 4 overlapped ranges per step (6 MB marks)
    assert line['config']['allreduce_calls_per_step'] >= 4, line['config']


def test_bench_py_four_ranks_share_one_gpu_over_gloo():
    """Rehearsal of the driver's `python bench.py --gpus N` beyond two ranks.  A one-GPU box allows six processes on its card,
    so four ranks (+ the launcher, which never touches the GPU) are what fits: world size, the overlapped all-reduce calls and — the property the
    scaling run depends on — identical parameters on every rank after three SUM-all-reduced steps.  The eight-rank case itself is the driver's to
    start; tests/test_parallel_gloo.py runs the collective layer with eight CPU ranks."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    p = subprocess.run([sys.executable, 'bench.py', '--gpus', '4', '--steps', '3', '--warmup', '1', '--batch', '2', '--variant', 's', '--size', '128',
                        '--no-roofline', '--no-cpu-baseline', '--no-infer'], cwd=ROOT, env=dict(env, YOLOv5_VERBOSE='false', OMP_NUM_THREADS='2'),
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith('{')][-1])
    cfg = line['config']
    assert line['n_gpus'] == 4 and cfg['world_size'] == 4 and cfg['global_batch'] == 8 and cfg['parallelism'] == 'dp4'
    assert cfg['allreduce_calls_per_step'] >= 4, cfg
    assert cfg['param_checksum_spread_over_ranks'] == 0.0, cfg
    assert line['value'] > 0 and line['final_loss'] == line['final_loss']


def test_train_py_cfg_hyp_freeze_masks_save_period(tmp_path):
    import yaml
    sys.path.insert(0, ROOT)
    from hd_yolo_amd import synth
    cfg = synth.make_cfg('n', 3)
    cfg['headers'][0][3][3] = 1                      # mask branch, one shared mask class (yolov5l6-mask.yaml:64)
    with open(tmp_path / 'model.yaml', 'w') as f:
        yaml.safe_dump(cfg, f)
    with open(tmp_path / 'hyp.yaml', 'w') as f:
        yaml.safe_dump({'lr0': 0.005, 'det': {'box': 0.04}}, f)
    out = run([sys.executable, 'train.py', '--cfg', str(tmp_path / 'model.yaml'), '--hyp', str(tmp_path / 'hyp.yaml'), '--freeze', 'backbone',
               '--masks', '--label-smoothing', '0.1', '--save-period', '1', '--batch-size', '4', '--imgsz', '128', '--epochs', '2',
               '--steps-per-epoch', '3', '--val-batches', '1', '--project', str(tmp_path), '--name', 'run', '--exist-ok', '--log-every', '1'])
    assert 'epochs completed' in out and 'det/mask' in out
    w = tmp_path / 'run' / 'weights'
    assert (w / 'epoch1.pt').exists() and (w / 'epoch2.pt').exists()
    a, b = torch.load(w / 'epoch1.pt', map_location='cpu'), torch.load(w / 'epoch2.pt', map_location='cpu')
    assert a['epoch'] == 1 and b['epoch'] == 2
    assert torch.equal(a['model']['backbone.3.conv.weight'], b['model']['backbone.3.conv.weight'])            # frozen
    assert torch.equal(a['model']['backbone.3.bn.running_mean'], b['model']['backbone.3.bn.running_mean'])    # FrozenBatchNorm2d
    assert not torch.equal(a['model']['neck.0.conv.weight'], b['model']['neck.0.conv.weight'])
    assert not torch.equal(a['model']['headers.det.seg_h.maskrcnn_preds.mask_fcn_logits.weight'],
                           b['model']['headers.det.seg_h.maskrcnn_preds.mask_fcn_logits.weight'])              # the mask head trained
    hyp = yaml.safe_load(open(tmp_path / 'run' / 'hyp.yaml'))
    assert hyp['lr0'] == 0.005 and hyp['det']['box'] == 0.04 and hyp['det']['label_smoothing'] == 0.1 and hyp['det']['cls'] == 0.5


def test_train_py_reads_reference_style_checkpoint(tmp_path):
    """The reference saves pickled half-precision modules under 'model' / 'ema' and 'epoch' = finished epochs (train.py:530-540)."""
    sys.path.insert(0, ROOT)
    from hd_yolo_amd import synth
    from metayolo.models.yolo import Model
    m = Model(synth.make_cfg('n', 2), synth.make_hyp())
    m.load_state_dict(synth.synth_state_dict(synth.shapes_of(m), seed=3), strict=False)
    ema = Model(synth.make_cfg('n', 2), synth.make_hyp())
    ema.load_state_dict(synth.synth_state_dict(synth.shapes_of(ema), seed=4), strict=False)
    torch.save({'epoch': 1, 'best_fitness': 0.0, 'model': torch.nn.Module.half(m), 'ema': torch.nn.Module.half(ema), 'updates': 7, 'optimizer': None},
               tmp_path / 'ref.pt')
    out = run([sys.executable, 'train.py', '--variant', 'n', '--nc', '2', '--weights', str(tmp_path / 'ref.pt'), '--batch-size', '4', '--imgsz', '64',
               '--epochs', '1', '--steps-per-epoch', '2', '--val-batches', '1', '--project', str(tmp_path), '--name', 'r', '--exist-ok', '--noval'])
    assert 'Transferred' in out and 'epochs completed' in out
    n = len(m.state_dict()) - sum('anchor' in k for k in m.state_dict())
    assert f'Transferred {n}/' in out
    out = run([sys.executable, 'val_nuclei.py', '--variant', 'n', '--nc', '2', '--imgsz', '64', '--batch-size', '2', '--batches', '1',
               '--weights', str(tmp_path / 'ref.pt')])
    assert 'fitness' in out


def test_bench_py_single_rank_over_rccl():
    """The N > 1 path with the real collective library: one rank, backend nccl (= RCCL), process group + DataParallel + the overlapped
    bucketed all-reduce on the comm stream — what a one-GPU box can exercise of `python bench.py --gpus N`."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(HDY_FORCE_DIST='1', HDY_DIST_BACKEND='nccl', YOLOv5_VERBOSE='false', HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_PORT='29547')
    p = subprocess.run([sys.executable, 'bench.py', '--gpus', '1', '--steps', '3', '--warmup', '1', '--batch', '8', '--no-roofline', '--no-cpu-baseline',
                        '--no-infer'], cwd=ROOT, env=env, capture_output=True, text=True, timeout=500)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith('{')][-1])
    assert line['config']['backend'] == 'nccl' and line['config']['world_size'] == 1 and line['value'] > 0
    assert line['config'].get('allreduce_calls_per_step', 0) >= 1, line['config']          # the bucketed all-reduce really went through RCCL
    # the self-diagnosis an N > 1 line carries (round 6): how long the main stream stood waiting for the collectives in front of the optimizer (one rank:
    # nothing on the wire, so next to nothing), the fastest / slowest rank's own clock, and the hardware queues the HIP runtime was started with
    c = line['config']
    assert {'allreduce_exposed_ms', 'rank_step_ms_min', 'rank_step_ms_max', 'hw_queues', 'hw_queues_in_time'} <= set(c), sorted(c)
    print('allreduce_exposed_ms', c['allreduce_exposed_ms'], 'rank step ms', c['rank_step_ms_min'], c['rank_step_ms_max'], 'queues', c['hw_queues'])
    assert c['allreduce_exposed_ms'] is not None and 0.0 <= c['allreduce_exposed_ms'] <= 0.3, c['allreduce_exposed_ms']
    assert c['rank_step_ms_min'] <= c['rank_step_ms_max'] == line['ms_per_step'] and c['hw_queues'] >= 8 and c['hw_queues_in_time'] is True
    assert line['steady_state'] and line['steady_state']['ms_per_step'] > 0 and line['prewarm_steps'] == 0


def _bench_line(env_extra, *flags):
    import json
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'HDY_FORCE_DIST')}
    env.update(YOLOv5_VERBOSE='false', HSA_ENABLE_IPC_MODE_LEGACY='0', **env_extra)
    p = subprocess.run([sys.executable, 'bench.py', '--gpus', '1', '--no-roofline', '--no-cpu-baseline', '--no-infer'] + list(flags), cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    return json.loads([l for l in p.stdout.splitlines() if l.startswith('{')][-1])


# allowed slow-down of the bench-size step when its gradients go through RCCL's all-reduce on the communication stream (one rank: the wire time is zero,
# what is left is the cost of the overlap machinery itself — stream waits at the marks, RCCL's kernels and queues beside the two compute streams)
RCCL_OVERLAP_MAX_RATIO = float(os.environ.get('HDY_RCCL_OVERLAP_MAX_RATIO', '1.04'))    # measured 1.021 (12.27 against 12.01 / 12.05 ms); 1.03 is the aim, one more point for box noise


def test_bench_size_step_with_overlapped_rccl_allreduce_costs_a_few_percent():
    """BASELINE configs[1] (yolov5s, batch 64, 640x640) with HDY_FORCE_DIST=1 — process group over RCCL, DataParallel, bucket marks, the communication
    stream — against the same step without any of it, on the same box, plain / RCCL / plain.  DESIGN.md §7's finding (HIP's default 4 hardware queues
    serialise the two launch lists once RCCL's streams exist: 14.69 against 13.82 ms) is what this pins: the package sets GPU_MAX_HW_QUEUES=8.  Also the
    self-check the first multi-GPU run relies on: measured calls and bytes per step equal the plan's `allreduce_expected`."""
    flags = ('--steps', '30', '--warmup', '5')
    a = _bench_line({}, *flags)
    d = _bench_line(dict(HDY_FORCE_DIST='1', HDY_DIST_BACKEND='nccl', MASTER_PORT='29549'), *flags)
    b = _bench_line({}, *flags)
    cfg = d['config']
    assert cfg['backend'] == 'nccl' and cfg['world_size'] == 1
    exp = cfg['allreduce_expected']
    assert exp['calls_per_step'] >= 3 and abs(cfg['allreduce_calls_per_step'] - exp['calls_per_step']) < 1e-6, cfg
    assert abs(cfg['allreduce_mb_per_step'] - exp['mb_per_step']) < 0.02 and abs(exp['mb_per_step'] - 7041205 * 4 / 2**20) < 0.1, cfg
    assert a['config']['allreduce_expected'] == exp                                  # the plan's marks do not depend on the process group
    plain = min(a['ms_per_step'], b['ms_per_step'])
    print(f'bench-size step: plain {a["ms_per_step"]} / {b["ms_per_step"]} ms, over RCCL (1 rank, overlapped) {d["ms_per_step"]} ms, ratio {d["ms_per_step"] / plain:.3f}')
    assert d['ms_per_step'] <= RCCL_OVERLAP_MAX_RATIO * plain, (a['ms_per_step'], d['ms_per_step'], b['ms_per_step'])
