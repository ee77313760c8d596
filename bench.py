#!/usr/bin/env python3
"""Headline benchmark: tiles/s (640x640) forward + backward + optimizer step of the metayolo detector on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of synthetic tiles already resident in HBM: HIP-plan forward of
backbone + neck + detection convs (bf16 operands, fp32 accumulate), target assignment + DetLoss, HIP-plan backward (BN/SiLU
backward, wgrad, dgrad), the RCCL gradient all-reduce when N > 1, and the SGD(Nesterov) update.  Workload = BASELINE.json
configs[1]: yolov5s, 8 classes, batch 64 per GPU, 640x640 (weak scaling: per-GPU batch fixed).  One JSON line on rank 0.

Extra objects on the line:
  roofline      the kernel on the layer the north star names — the fused 3x3 conv 64->64 at 80x80, batch 64 (filter-resident
                conv3x3_c64) — timed live with HIP events on the launch stream; achieved = 2*N*K*C*9*Ho*Wo / t.  Its `layers_3x3`
                lists EVERY 3x3 convolution launch of the benchmarked model (forward and data gradient) timed alone, `min_frac` is the
                worst of them: the named kernel is 2 % of the step, the others are where the time goes.  Each row also names the
                roofline that bounds THAT layer (`bound`: max(FLOPs / 2.5 PFLOP/s, read-once/write-once bytes / 8 TB/s)) and its
                `frac_of_bound`: the stride-2 layers at 320x320 / 160x160 are HBM-bound (0.24 of the MFMA peak AT the HBM roofline).
  roofline_hbm  the step's dominant HBM-bound kernel (BatchNorm backward over the largest activation) timed alone, bytes / time vs 8 TB/s;
                `all_launches`: the same over every BatchNorm-backward call of the plan (the kernel family with the most time in the step).
  step          whole-step fractions: conv FLOPs of the step / time / 2.5 PFLOP/s and read-once/write-once bytes of its conv and
                BatchNorm launches / time / 8 TB/s.
  infer         BASELINE configs[3] (yolov5l, batch 128, 1024x1024 inference): tiles/s, decode and NMS microseconds per tile.
  cpu_baseline  the CPU oracle (oracle/ref_net.py, a torch-fp32 port of the reference path) doing the same training step on
                a bounded sample (a few 640x640 tiles) on this host's cores.
  steady_state  a SECOND block of K timed steps right after the headline block (same fences): what the step costs once clocks and caches are warm.
                `value` / `ms_per_step` are always the first block — W warm-up steps, then exactly K timed steps, no pre-warm (HDY_BENCH_PREWARM_S opts in).
  config        also carries the self-diagnosis of an N > 1 run: allreduce_exposed_ms (main stream waiting for the collectives in front of the
                optimizer), rank_step_ms_min / max, hw_queues / hw_queues_in_time, allreduce_expected against allreduce_calls_per_step / _mb_per_step.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault('YOLOv5_VERBOSE', 'false')
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: RCCL across processes needs it on this driver
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')               # side stream + comm stream + RCCL's own: see hd_yolo_amd/__init__.py
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def _requested_gpus(argv):
    for i, a in enumerate(argv):
        if a == '--gpus' and i + 1 < len(argv):
            return int(argv[i + 1])
        if a.startswith('--gpus='):
            return int(a.split('=', 1)[1])
    return 1


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a rendezvous in the environment: start N fresh rank processes (one per GPU) under
    torch.distributed.run — before this process has imported torch or touched the GPU — relay their output and exit with their
    code.  Rank 0's JSON line is the only thing they print on stdout."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=dict(os.environ, HDY_BENCH_SPAWNED='1'))


if __name__ == '__main__' and 'RANK' not in os.environ and _requested_gpus(sys.argv[1:]) > 1:
    sys.exit(spawn_ranks(_requested_gpus(sys.argv[1:]), sys.argv[1:]))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from hd_yolo_amd import ops, synth  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA peak, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def make_optimizer(model, hyp, batch_total):
    """Three parameter groups and scaled weight decay as train.py:208-233."""
    nbs = 64
    accumulate = max(round(nbs / batch_total), 1)
    wd = hyp['weight_decay'] * batch_total * accumulate / nbs
    g_bn, g_w, g_b = [], [], []
    for m in model.modules():
        if hasattr(m, 'bias') and isinstance(m.bias, torch.nn.Parameter):
            g_b.append(m.bias)
        if isinstance(m, torch.nn.BatchNorm2d):
            g_bn.append(m.weight)
        elif hasattr(m, 'weight') and isinstance(m.weight, torch.nn.Parameter):
            g_w.append(m.weight)
    from hd_yolo_amd.optim import SGD          # torch.optim.SGD's update, one launch for all tensors (csrc/optim.hip)
    opt = SGD(g_bn, lr=hyp['lr0'], momentum=hyp['momentum'], nesterov=True)
    opt.add_param_group({'params': g_w, 'weight_decay': wd})
    opt.add_param_group({'params': g_b})
    return opt


def pmc_traffic():
    """HBM bytes per launch of the roofline kernel from the committed rocprofv3 --pmc passes (FETCH_SIZE x2 + WRITE_SIZE, see
    profiles/r03_conv3x3_pmc.json and scripts/roofline_kernel.py); None when the file is absent."""
    for name in ('r06_conv3x3_pmc.json', 'r05_conv3x3_pmc.json', 'r04_conv3x3_pmc.json', 'r03_conv3x3_pmc.json'):
        try:
            with open(os.path.join(ROOT, 'profiles', name)) as f:
                return json.load(f)['traffic_bytes_per_launch']
        except (OSError, KeyError, ValueError):
            continue
    return None


def conv_roofline(device, iters=30):
    """Time the bf16 3x3 conv kernel on yolov5s' 64->64 @ 80x80 layer at batch 64 (30.2 GFLOP per launch), through the C ABI."""
    N, H, W, C, K = 64, 80, 80, 64, 64
    dt = torch.bfloat16
    x = torch.randn((N, H, W, C), device=device).to(dt)
    w = torch.randn((K, C, 3, 3), device=device) * 0.05
    y = torch.empty((N, H, W, K), dtype=dt, device=device)
    wp = ops.pack_alloc(K, C, 3, 3, 1, 1, ops.PACK_FWD, dt, device)
    mt = ops.stat_slabs(N, H, W, C, K, 3, 3, 1, 1, dt)
    stats = torch.empty((mt, 2, K), dtype=torch.float32, device=device)
    ops.run([ops.rec_pack(w, None, 1, 1, ops.PACK_FWD, wp)])
    rec = [ops.rec_conv_fwd(x, wp, y, K, 3, 3, 1, 1, stats=stats)]
    for _ in range(5):
        ops.run(rec)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        ops.run(rec)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / iters * 1e-3
    flops = 2.0 * N * K * C * 9 * H * W
    ach = flops / t / 1e12
    return {'bound': 'mfma', 'achieved': round(ach, 2), 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(ach / PEAK_BF16_TFLOPS, 4),
            'traffic': pmc_traffic(), 'kernel': 'conv3x3_c64_kernel (bf16, filter in registers) fwd 3x3 64->64 @80x80 B=64 (+BN stat slabs)',
            'us_per_launch': round(t * 1e6, 2), 'algorithmic_gflop_per_launch': round(flops / 1e9, 2)}


def cpu_baseline(variant, nc, size, tiles=8, iters=16):
    """Reference-path port on the host: oracle train step (forward + DetLoss + backward) on `tiles` 640x640 tiles."""
    from oracle.ref_net import RefNet
    from hd_yolo_amd import host_cpu_quota
    torch.set_num_threads(host_cpu_quota())                      # the cores the cgroup really grants (16 on a 1-GPU box)
    net = RefNet(synth.make_cfg(variant, nc), synth.make_hyp())
    sd = net.init_state()
    for k, t in sd.items():
        if 'running' not in k:
            t.requires_grad_(True)
    x = synth.synth_images(tiles, size, seed=0)
    targets = synth.synth_targets(tiles, size, nc, seed=1)
    times = []
    for i in range(iters + 1):
        t0 = time.perf_counter()
        loss, _, _ = net.train_forward(sd, x, targets)
        loss.backward()
        times.append(time.perf_counter() - t0)
        for t in sd.values():
            t.grad = None
    t = sorted(times[1:])[len(times[1:]) // 2]
    return {'value': round(tiles / t, 3), 'unit': 'tiles/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'oracle/ref_net.py (torch fp32 CPU restatement of the reference path), yolov5{variant} nc={nc}, '
                      f'{tiles} tiles {size}x{size}, fwd+DetLoss+bwd, median of {iters} after 1 warm-up'}


def cpu_nms_baseline(preds, nc, conf_thres, iou_thres, max_det):
    """CPU leg of BASELINE configs[3] ("decode+NMS HIP path vs CPU torchvision.nms"): oracle/nms_ref.c — the scalar C restatement of the filter +
    torchvision greedy NMS the reference calls (utils_general.py:299-356) — on a sample of the tiles the HIP kernel just processed (same decoded
    candidates, same threshold).  One thread: the reference's per-image loop is serial too."""
    from oracle import nms_ref
    nms_ref.build()
    nms_ref.nms_batched_c(preds[:1], nc, conf_thres, iou_thres, max_det)                  # load the library, touch the pages
    t0 = time.perf_counter()
    _, n_keep, _ = nms_ref.nms_batched_c(preds, nc, conf_thres, iou_thres, max_det)
    dt = time.perf_counter() - t0
    return {'us_per_tile': round(dt / len(preds) * 1e6, 1), 'cores': 1, 'kind': 'port', 'tiles': int(len(preds)), 'kept_per_tile': round(float(n_keep.mean()), 1),
            'sample': f'oracle/nms_ref.c (filter + greedy NMS, scalar C), the first {len(preds)} of the batch\'s decoded tiles, {preds.shape[1]} candidates each'}


def reference_c1():
    """The reference's OWN code timed on BASELINE configs[0] (yolov5n, 2 classes, batch 4, 640x640, fp32, CPU) by scripts/time_reference_cpu.py in the
    build container — /root/reference cannot travel to the GPU box, its JSON summary does."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'ref_cpu_c1.json')) as f:
            r = json.load(f)
        return {k: r[k] for k in ('train_tiles_per_s', 'eval_tiles_per_s', 'threads', 'torch', 'host') if k in r}
    except (OSError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=60)
    ap.add_argument('--warmup', type=int, default=15)
    ap.add_argument('--batch', type=int, default=64, help='tiles per GPU')
    ap.add_argument('--size', type=int, default=640)
    ap.add_argument('--variant', default='s')
    ap.add_argument('--nc', type=int, default=8)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-infer', action='store_true', help='skip the C4 inference measurement (yolov5l, batch 128, 1024x1024)')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        # the product path has no CPU fallback; under `--gpus N` this is the failing rank whose exit code spawn_ranks relays
        print(f'bench.py rank {rank}: no MI355X visible (torch.cuda.is_available() is False) — hd_yolo_amd has no CPU path', file=sys.stderr, flush=True)
        sys.exit(3)
    # HDY_FORCE_DIST=1: go through process-group init, DataParallel and the overlapped all-reduce with ONE rank too — the only way to run
    # RCCL itself on a one-GPU box (rehearsal of the N > 1 path, tests/test_gpu_entrypoints.py)
    force_dist = os.environ.get('HDY_FORCE_DIST') == '1'
    if world > 1 or force_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29541')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        torch.cuda.set_device(local % torch.cuda.device_count())
        # 'gloo' lets several ranks share one GPU when rehearsing the N>1 path; it is also what a box with fewer GPUs than ranks gets
        backend = os.environ.get('HDY_DIST_BACKEND', 'nccl' if torch.cuda.device_count() >= world else 'gloo')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend)
    else:
        backend = 'none'
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: the launcher and the flag disagree')
    device = torch.device('cuda', local % torch.cuda.device_count())
    torch.cuda.set_device(device)

    from metayolo.models.yolo import Model
    from hd_yolo_amd.parallel import DataParallel
    hyp = synth.make_hyp()
    # the reference ships no hyp file; these are the YOLOv5 scratch defaults except warmup_bias_lr: with 50-400 synthetic
    # boxes per tile and random weights, a 0.1 bias learning rate drives box logits to sigmoid saturation within ~25 steps
    # (NaN in CIoU's aspect term, in the reference's autograd as well), so biases warm up from 0 like the weights
    hyp['warmup_bias_lr'] = 0.0
    model = Model(synth.make_cfg(args.variant, args.nc), hyp)
    model.load_state_dict(synth.synth_state_dict(synth.shapes_of(model), seed=0), strict=False)
    model = model.to(device).train()
    if args.dtype == 'bf16':
        model.half()                      # bf16 operands, fp32 accumulate / master weights
    net = DataParallel(model) if (world > 1 or force_dist) else model
    opt = make_optimizer(model, hyp, args.batch * world)

    x = synth.synth_images(args.batch, args.size, seed=rank).to(device)
    targets = synth.synth_targets(args.batch, args.size, args.nc, seed=1 + rank)
    for t in targets:
        for a in t['anns']['det']:
            a['boxes'], a['labels'] = a['boxes'].to(device), a['labels'].to(device)

    # learning-rate / momentum warm-up exactly as train.py:354,436-444 (nw = max(3 epochs, 100 iterations)): weights ramp up
    # from 0, biases down from warmup_bias_lr.  A 64-tile synthetic "epoch" is one iteration, so nw = 100.
    nw, it = 100, [0]
    lf = lambda e: (1 - e / 300) * (1.0 - hyp['lrf']) + hyp['lrf']          # linear schedule, epoch 0

    def warm():
        ni = it[0]
        if ni <= nw:
            for j, g in enumerate(opt.param_groups):
                lo = hyp['warmup_bias_lr'] if j == 2 else 0.0
                g['lr'] = lo + (hyp['lr0'] * lf(0) - lo) * ni / nw
                g['momentum'] = hyp['warmup_momentum'] + (hyp['momentum'] - hyp['warmup_momentum']) * ni / nw
        it[0] += 1

    nsteps_run = [0]

    def step():
        nsteps_run[0] += 1
        warm()
        losses, _ = net(x, targets, compute_masks=False)
        loss = losses['det']['det_loss']
        loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        return loss

    def fence():
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # Pre-warm is OPT-IN (HDY_BENCH_PREWARM_S=<seconds>; round 5 ran it by default and the advisor was right that the headline then was not the
    # driver's `--warmup W` measurement: with lr = 0 the SGD kernel still integrates the momentum buffers and BatchNorm's running statistics
    # move).  When asked for, the untimed steps run with the schedule held at iteration 0 and the optimizer's momentum buffers and the BatchNorm
    # running statistics are put back afterwards, so the counted warm-up starts from the state it would have had; `prewarm_steps` says how many
    # ran.  The default line is W warm-up steps, then EXACTLY K timed steps (`value`), then a SECOND block of K timed steps reported beside it as
    # `steady_state` (same fences; by then the clocks and caches are warm) — both numbers, never only the better one.
    prewarm_target = float(os.environ.get('HDY_BENCH_PREWARM_S', '0'))
    prewarm_s, prewarm_steps = 0.0, 0
    if prewarm_target > 0:
        keep_buf = {k: v.detach().clone() for k, v in model.state_dict().items() if 'running_' in k or 'num_batches_tracked' in k}
        step()                                    # builds the plans, loads the code objects: not counted towards the target
        it[0] = 0
        torch.cuda.synchronize()
        while prewarm_s < prewarm_target and prewarm_steps < 2000:
            tp = time.perf_counter()
            for _ in range(10):
                step()
                it[0] = 0
            torch.cuda.synchronize()
            dtp = time.perf_counter() - tp
            if world > 1 or force_dist:
                # every rank must run the SAME number of steps (each step holds collectives): the slowest rank's clock decides for all
                tblk = torch.tensor([dtp], device=device, dtype=torch.float64)
                dist.all_reduce(tblk, op=dist.ReduceOp.MAX)
                dtp = tblk.item()
            prewarm_s += dtp
            prewarm_steps += 10
        with torch.no_grad():                     # back to the state of a process that never pre-warmed: BatchNorm buffers as loaded, no momentum buffers
            sd_now = model.state_dict()
            for k, v in keep_buf.items():
                sd_now[k].copy_(v)
            opt.state.clear()
            opt._key = None
    for _ in range(args.warmup):
        step()
    fence()
    if os.environ.get('HDY_BENCH_PHASES'):          # diagnostic only: serialises the phases, never used for the reported value
        acc = [0.0, 0.0, 0.0]
        for _ in range(5):
            warm()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            losses, _ = net(x, targets, compute_masks=False)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            losses['det']['det_loss'].backward()
            torch.cuda.synchronize(); t2 = time.perf_counter()
            opt.step(); opt.zero_grad(set_to_none=True)
            torch.cuda.synchronize(); t3 = time.perf_counter()
            acc = [acc[0] + t1 - t0, acc[1] + t2 - t1, acc[2] + t3 - t2]
        if rank == 0:
            print('phases ms (fwd+loss, bwd, opt):', [round(a / 5 * 1e3, 2) for a in acc], file=sys.stderr, flush=True)
    dp_on = world > 1 or force_dist
    if dp_on:
        net.reducer.exposed_events = []             # two events around the main stream's wait for the communication stream, every step from here on
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    dt_min = dt
    if dp_on:
        tmax = torch.tensor([dt], device=device, dtype=torch.float64)
        tmin = tmax.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        dt, dt_min = tmax.item(), tmin.item()
    # time the main stream stood waiting for the all-reduce in front of the optimizer (0 = every collective finished under the backward pass): this
    # rank's average over the timed steps, then the largest over the ranks
    exposed_ms = None
    if dp_on:
        ev = net.reducer.exposed_events
        net.reducer.exposed_events = None
        mine = sum(a.elapsed_time(b) for a, b in ev) / max(len(ev), 1)
        tex = torch.tensor([mine], device=device, dtype=torch.float64)
        dist.all_reduce(tex, op=dist.ReduceOp.MAX)
        exposed_ms = tex.item()
    # second block of K steps, same fences: the steady-state number beside the headline (never instead of it).  HDY_BENCH_SECOND_BLOCK=0 (the
    # profiling scripts, which count kernels per step over warm-up + K steps) skips it.
    second = os.environ.get('HDY_BENCH_SECOND_BLOCK', '1') != '0'
    t0 = time.perf_counter()
    for _ in range(args.steps if second else 0):
        step()
    fence()
    dt2 = time.perf_counter() - t0
    if dp_on:
        tmax = torch.tensor([dt2], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt2 = tmax.item()
    final_loss = float(loss.detach())
    param_spread = 0.0
    if world > 1 or force_dist:
        # every rank must hold the same parameters after the same number of SUM-all-reduced steps: largest minus smallest per-rank checksum
        chk = torch.stack([p.detach().double().sum() for p in model.parameters()]).sum().reshape(1)
        hi, lo = chk.clone(), chk.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        param_spread = float(hi - lo)

    if rank == 0:
        tiles = args.batch * world * args.steps
        line = {
            'metric': 'tiles/sec (640x640) fwd+bwd', 'value': round(tiles / dt, 2), 'unit': 'tiles/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16' if args.dtype == 'bf16' else 'f32', 'data': 'synthetic',
            'config': {'workload': f'metayolo yolov5{args.variant} {args.nc}-class nuclei, {args.size}x{args.size} RGB tiles, '
                                   f'batch {args.batch}/GPU, train step = fwd + DetLoss + bwd + all-reduce + SGD(nesterov)',
                       'global_batch': args.batch * world, 'parallelism': f'dp{world}',
                       'world_size': dist.get_world_size() if (world > 1 or force_dist) else 1, 'backend': backend,
                       'allreduce_calls_per_step': (round(net.reducer.calls / max(nsteps_run[0], 1), 2) if (world > 1 or force_dist) else 0),
                       'param_checksum_spread_over_ranks': param_spread},
            'final_loss': round(final_loss, 4), 'prewarm_s': round(prewarm_s, 3), 'prewarm_steps': prewarm_steps,
            'steady_state': ({'ms_per_step': round(dt2 / args.steps * 1e3, 3), 'value': round(tiles / dt2, 2),
                              'what': f'a second block of {args.steps} timed steps right after the headline block (same barrier + synchronize on both sides)'}
                             if second else None),
        }
        import hd_yolo_amd
        # self-diagnosis of an N > 1 run: where a scaling loss would come from.  rank_step_ms_min / max: the fastest and the slowest rank's own clock
        # over the timed block (`ms_per_step` is the max); allreduce_exposed_ms: see above; hw_queues: GPU_MAX_HW_QUEUES as the HIP runtime of this
        # process read it (the package sets 8 before the runtime loads; `hw_queues_in_time` False = it was already up and the default 4 applies)
        line['config'].update(rank_step_ms_min=round(dt_min / args.steps * 1e3, 3), rank_step_ms_max=round(dt / args.steps * 1e3, 3),
                              allreduce_exposed_ms=(None if exposed_ms is None else round(exposed_ms, 4)),
                              hw_queues=int(os.environ.get('GPU_MAX_HW_QUEUES', '4')), hw_queues_in_time=bool(hd_yolo_amd.HW_QUEUES_IN_TIME))
        from hd_yolo_amd import bench_util
        from hd_yolo_amd.parallel import GradAllReduce
        plan = next(iter(model._eng().plans.values()))
        # what one backward pass hands to RCCL when N > 1 (from the plan's "range is final" marks, hd_yolo_amd/parallel.py): printed at every N so that
        # the first multi-GPU run checks itself — `allreduce_calls_per_step` (measured) must equal `calls_per_step` here and the bytes must add up to 4 x parameters
        sends = GradAllReduce.expected_sends(plan.bucket_marks(), plan.grad_store.numel)
        line['config']['allreduce_expected'] = {'calls_per_step': len(sends), 'mb_per_call': [round((b - a) * 4 / 2**20, 2) for a, b in sends],
                                                'mb_per_step': round(sum(b - a for a, b in sends) * 4 / 2**20, 2)}
        if world > 1 or force_dist:
            line['config']['allreduce_mb_per_step'] = round(net.reducer.bytes / max(nsteps_run[0], 1) / 2**20, 2)
        fl, by = bench_util.plan_work(plan)
        ms = dt / args.steps * 1e3
        line['step'] = {'conv_tflop_per_step': round(fl / 1e12, 3), 'mfma_frac': round(fl / ms / 1e9 / PEAK_BF16_TFLOPS, 4),
                        'algorithmic_gb_per_step': round(by / 1e9, 2), 'hbm_frac': round(by / ms / 1e6 / PEAK_HBM_GBS, 4),
                        'launches_per_step': len(bench_util.flat_records(plan.fwd)) + len(bench_util.flat_records(plan.bwd))}
        # measured HBM bytes of the whole step (rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE over 13 steps, scripts/step_traffic.sh) against the
        # algorithmic bytes: well above 1 = wasted re-reads.  From the committed summary of the same command (PMC passes cannot run inside the timed process).
        pdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles')
        tj = next((n for n in ('r06_step_traffic.json', 'r05_step_traffic.json', 'r04_step_traffic.json', 'r03_step_traffic.json') if os.path.exists(os.path.join(pdir, n))), None)
        if args.variant == 's' and args.batch == 64 and args.size == 640 and tj:
            tr = json.load(open(os.path.join(pdir, tj)))
            line['step'].update(traffic_gb_per_step=tr['hbm_gb_per_step'], traffic_ratio=round(tr['hbm_gb_per_step'] / (by / 1e9), 3),
                                traffic_source='profiles/' + tj)
        if not args.no_roofline:
            line['roofline'] = conv_roofline(device)
            rows = bench_util.conv3x3_table(plan, PEAK_BF16_TFLOPS)
            line['roofline']['layers_3x3'] = rows
            line['roofline']['min_frac'] = min(r['frac'] for r in rows)
            line['roofline']['min_frac_of_bound'] = min(r['frac_of_bound'] for r in rows)
            line['roofline_hbm'] = bench_util.hbm_kernel_roofline(plan, PEAK_HBM_GBS)
        if world == 1 and not args.no_infer and args.variant == 's' and args.batch == 64:
            try:
                line['infer'] = bench_util.infer_benchmark('l', 128, 1024, 3, device, cpu_nms=None if args.no_cpu_baseline else cpu_nms_baseline)
            except Exception as e:                                     # never lose the headline line to the side measurement
                line['infer'] = {'error': repr(e)[:200]}
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(args.variant, args.nc, args.size)
            ref = reference_c1()
            if ref is not None:
                line['cpu_baseline']['reference_c1'] = ref
                line['cpu_baseline']['sample'] += ('; reference_c1 = the reference\'s own train.py:455-472 loop body / val forward on configs[0], timed by '
                                                   'scripts/time_reference_cpu.py (profiles/ref_cpu_c1.json), other host')
        print(json.dumps(line), flush=True)
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
