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
    assert ck['epoch'] == 1 and 'backbone.0.conv.weight' in ck['model'] and 'headers.det.m.2.bias' in ck['ema']
    assert all(torch.isfinite(v).all() for v in ck['model'].values() if v.dtype.is_floating_point)
    out = run(base + ['--epochs', '3', '--resume', '--weights', str(tmp_path / 'run' / 'weights' / 'last.pt')])
    assert 'epoch 2/2' in out and 'epoch 0/2' not in out
    out = run([sys.executable, 'val_nuclei.py', '--variant', 'n', '--nc', '2', '--imgsz', '128', '--batch-size', '4', '--batches', '2',
               '--weights', str(tmp_path / 'run' / 'weights' / 'last.pt')])
    assert 'fitness' in out


def test_train_py_two_ranks_gloo(tmp_path):
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29531', 'train.py', '--variant', 'n', '--nc', '2', '--batch-size', '8', '--imgsz', '64', '--epochs', '1',
           '--steps-per-epoch', '3', '--val-batches', '1', '--project', str(tmp_path), '--name', 'dp', '--exist-ok']
    out = run(cmd, env={'HDY_DIST_BACKEND': 'gloo'})
    assert 'epochs completed' in out
