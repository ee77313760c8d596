#!/usr/bin/env python3
"""Time the REFERENCE's own code on BASELINE.json configs[0] (C1): metayolo yolov5n, 2 classes, batch 4, synthetic 640x640 tiles, fp32,
PyTorch CPU.  Build container only: it imports /root/reference (under tests/golden/make_golden.install_shims(), the same import the
golden fixtures are generated with) — the GPU box never sees the reference, only the JSON this script writes:

    python scripts/time_reference_cpu.py            ->  profiles/ref_cpu_c1.json

What is timed:
  train   the body of the reference's loop, /root/reference/train.py:455-472: `losses, _ = model(imgs, targets, compute_masks=True)`,
          the loss sum over the headers, `.backward()`, and the optimizer step of :475-479 with torch.optim.SGD as train.py:236 builds it
          (no AMP on CPU: `amp.autocast(enabled=cuda)` is a no-op there); 1 warm-up + ITERS timed iterations, median.
  eval    `model(imgs)` in eval mode as val_nuclei.py:143 calls it; its NMS is the stand-in for the absent torchvision (the oracle's numpy
          restatement), so the eval figure is labelled with that.
Threads = the cores this container grants (os.cpu_count(), the cgroup quota if lower); torch version and parallel backend are recorded.
"""
import importlib.util
import json
import os
import platform
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location('make_golden', os.path.join(ROOT, 'tests', 'golden', 'make_golden.py'))
mg = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mg)

from hd_yolo_amd import host_cpu_quota, synth  # noqa: E402

VARIANT, NC, BATCH, SIZE = 'n', 2, 4, 640
ITERS = int(os.environ.get('ITERS', '15'))


def median(v):
    v = sorted(v)
    return v[len(v) // 2]


def main():
    assert os.path.isdir(mg.REF), 'the reference is only mounted in the build container'
    threads = host_cpu_quota()
    torch.set_num_threads(threads)
    mg.install_shims()
    hyp = synth.make_hyp()
    model = mg.ref_model(VARIANT, NC, hyp).train()
    x = synth.synth_images(BATCH, SIZE, seed=0)
    targets = synth.synth_targets(BATCH, SIZE, NC, seed=1)
    # train.py:208-236: three parameter groups, SGD with Nesterov momentum
    g_bn, g_w, g_b = [], [], []
    for m in model.modules():
        if hasattr(m, 'bias') and isinstance(m.bias, torch.nn.Parameter):
            g_b.append(m.bias)
        if isinstance(m, torch.nn.BatchNorm2d):
            g_bn.append(m.weight)
        elif hasattr(m, 'weight') and isinstance(m.weight, torch.nn.Parameter):
            g_w.append(m.weight)
    opt = torch.optim.SGD(g_bn, lr=0.0, momentum=hyp['momentum'], nesterov=True)      # lr 0: the weights stay the synthetic ones (timing only)
    opt.add_param_group({'params': g_w, 'weight_decay': hyp['weight_decay']})
    opt.add_param_group({'params': g_b})

    t_train = []
    for i in range(ITERS + 1):
        t0 = time.perf_counter()
        losses, _ = model(x, targets, compute_masks=True)
        loss = 0.
        for task_losses in losses.values():
            loss = loss + task_losses['det_loss']
        loss.backward()
        opt.step()
        opt.zero_grad()
        t_train.append(time.perf_counter() - t0)
    final_loss = float(loss.detach().reshape(-1)[0])

    model.eval()
    t_eval = []
    with torch.no_grad():
        for i in range(ITERS + 1):
            t0 = time.perf_counter()
            _, outs = model(x)
            t_eval.append(time.perf_counter() - t0)
    tt, te = median(t_train[1:]), median(t_eval[1:])
    out = {
        'config': f'BASELINE configs[0]: metayolo yolov5{VARIANT}, {NC} classes, batch {BATCH}, synthetic {SIZE}x{SIZE} tiles, fp32, reference PyTorch CPU path',
        'what': 'the reference\'s own modules (/root/reference/metayolo/models, imported under tests/golden/make_golden.install_shims()): '
                'train = train.py:455-479 loop body (forward + DetLoss + backward + SGD step), eval = val_nuclei.py:143 forward incl. NMS '
                '(stand-in for the absent torchvision.ops.nms: oracle/nms_ref.nms_numpy)',
        'train_s_per_iter': round(tt, 4), 'train_tiles_per_s': round(BATCH / tt, 2), 'train_tiles_per_s_best_iter': round(BATCH / min(t_train[1:]), 2),
        'train_iters_s': [round(t, 4) for t in t_train],
        'eval_s_per_iter': round(te, 4), 'eval_tiles_per_s': round(BATCH / te, 2), 'eval_tiles_per_s_best_iter': round(BATCH / min(t_eval[1:]), 2),
        'eval_iters_s': [round(t, 4) for t in t_eval],
        'iters': ITERS, 'warmup': 1, 'statistic': 'median', 'final_loss': round(final_loss, 5),
        'detections_per_tile': [int(len(o['det']['boxes'])) for o in outs],
        'threads': threads, 'cpu_count': os.cpu_count(), 'torch': torch.__version__, 'host': platform.processor() or platform.machine(),
        'parallel_info': torch.__config__.parallel_info().strip().splitlines()[:6],
    }
    try:
        with open('/proc/cpuinfo') as f:
            names = [l.split(':', 1)[1].strip() for l in f if l.startswith('model name')]
        if names:
            out['host'] = names[0]
    except OSError:
        pass
    dst = os.path.join(ROOT, 'profiles', 'ref_cpu_c1.json')
    with open(dst, 'w') as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: out[k] for k in ('train_tiles_per_s', 'train_tiles_per_s_best_iter', 'eval_tiles_per_s', 'eval_tiles_per_s_best_iter', 'threads', 'torch', 'host')}))


if __name__ == '__main__':
    main()
