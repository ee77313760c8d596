"""bench.py's launcher contract on a box without a GPU: `--gpus N` with no rendezvous in the environment starts its own ranks
(torch.distributed.run on 127.0.0.1, free port) BEFORE torch is imported, and a failing rank's exit status comes back as bench.py's own —
here every rank fails at once, loudly, because the product path has no CPU fallback."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_under_test', os.path.join(ROOT, 'bench.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)         # __name__ != '__main__': nothing is spawned or run
    return mod


def test_gpus_flag_is_parsed_before_torch_is_imported():
    b = _load_bench()
    assert b._requested_gpus(['--steps', '3']) == 1
    assert b._requested_gpus(['--gpus', '8', '--steps', '3']) == 8
    assert b._requested_gpus(['--steps', '3', '--gpus=4']) == 4


@pytest.mark.timeout(300)
def test_a_failing_rank_fails_the_launcher():
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is visible: the ranks would run (tests/test_gpu_entrypoints.py covers that side)')
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    p = subprocess.run([sys.executable, 'bench.py', '--gpus', '2', '--steps', '1', '--warmup', '0', '--no-roofline', '--no-cpu-baseline', '--no-infer'],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert p.returncode != 0, 'two ranks without a GPU must not report success'
    assert 'no MI355X visible' in p.stderr and 'rank 1' in p.stderr, p.stderr[-3000:]
    assert not [l for l in p.stdout.splitlines() if l.startswith('{')], 'no JSON line from a failed run'
