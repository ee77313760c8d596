#!/usr/bin/env python3
"""Training entry point with the reference's structure (reference: train.py:87-588 train(), :591-641 argument_parser,
:644-690 main) on the MI355X path, for synthetic tiles.  Kept from the reference because they change results:
nominal batch 64 / accumulate / scaled weight decay (:208-210), three parameter groups (:213-226), SGD-Nesterov | Adam |
AdamW (:229-233), linear or one-cycle LambdaLR (:242-246), warm-up of lr and momentum (:354, :436-444), EMA (:249),
per-header loss-gain rescale (:335-345), one process per GPU with env:// rendezvous (:67-69, :683).
Replaced: DistributedDataParallel -> hd_yolo_amd.parallel.DataParallel (flat-bucket RCCL sum all-reduce; the loss is NOT
multiplied by WORLD_SIZE, cf. :467); amp.autocast/GradScaler -> bf16 operands with fp32 masters (no loss scaling needed);
datasets/loggers/plots/evolve/W&B -> out of scope.

    python train.py --variant s --nc 8 --batch-size 64 --imgsz 640 --epochs 2 --steps-per-epoch 20
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train.py --batch-size 512 ...
"""
import argparse
import json
import math
import os
import sys
import time
from copy import deepcopy
from pathlib import Path

import torch
import torch.distributed as dist
from torch.optim import SGD, Adam, AdamW, lr_scheduler

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import val_nuclei as val  # noqa: E402
from hd_yolo_amd import synth  # noqa: E402
from hd_yolo_amd.parallel import DataParallel  # noqa: E402
from metayolo import LOGGER  # noqa: E402
from metayolo.common import ModelEMA, de_parallel  # noqa: E402
from metayolo.datasets import SyntheticTiles  # noqa: E402
from metayolo.engines.general import increment_path, init_seeds, one_cycle  # noqa: E402
from metayolo.engines.torch_utils import select_device  # noqa: E402
from metayolo.models.utils_general import check_img_size  # noqa: E402
from metayolo.models.utils_torch import EarlyStopping  # noqa: E402
from metayolo.models.yolo import Model  # noqa: E402

LOCAL_RANK = int(os.getenv('LOCAL_RANK', -1))
RANK = int(os.getenv('RANK', -1))
WORLD_SIZE = int(os.getenv('WORLD_SIZE', 1))


def build_optimizer(model, hyp, name):
    g_bn, g_w, g_b = [], [], []
    for m in model.modules():
        if hasattr(m, 'bias') and isinstance(m.bias, torch.nn.Parameter):
            g_b.append(m.bias)
        if isinstance(m, torch.nn.BatchNorm2d):
            g_bn.append(m.weight)
        elif hasattr(m, 'weight') and isinstance(m.weight, torch.nn.Parameter):
            g_w.append(m.weight)
    if name == 'Adam':
        opt = Adam(g_bn, lr=hyp['lr0'], betas=(hyp['momentum'], 0.999))
    elif name == 'AdamW':
        opt = AdamW(g_bn, lr=hyp['lr0'], betas=(hyp['momentum'], 0.999))
    else:
        opt = SGD(g_bn, lr=hyp['lr0'], momentum=hyp['momentum'], nesterov=True)
    opt.add_param_group({'params': g_w, 'weight_decay': hyp['weight_decay']})
    opt.add_param_group({'params': g_b})
    return opt


def train(hyp, opt, device):
    save_dir = Path(opt.save_dir)
    w = save_dir / 'weights'
    if RANK in (-1, 0):
        w.mkdir(parents=True, exist_ok=True)
        with open(save_dir / 'opt.json', 'w') as f:
            json.dump(vars(opt), f, indent=1, default=str)
    init_seeds(1 + RANK)
    cfg = synth.make_cfg(opt.variant, opt.nc)
    model = Model(cfg, hyp).to(device)
    start_epoch, best_fitness = 0, 0.0
    ckpt = None
    if opt.weights:
        ckpt = torch.load(opt.weights, map_location='cpu')
        sd = ckpt['model'] if isinstance(ckpt, dict) and 'model' in ckpt else ckpt
        sd = {k: v for k, v in sd.items() if k in model.state_dict() and v.shape == model.state_dict()[k].shape and 'anchor' not in k}
        model.load_state_dict(sd, strict=False)
        LOGGER.info(f'Transferred {len(sd)}/{len(model.state_dict())} items from {opt.weights}')
    gs = 32
    imgsz = check_img_size(opt.imgsz, gs, floor=gs * 2)
    batch_size = opt.batch_size // WORLD_SIZE                         # per-rank batch, as train.py:287

    nbs = 64
    accumulate = max(round(nbs / opt.batch_size), 1)
    hyp['weight_decay'] *= opt.batch_size * accumulate / nbs
    optimizer = build_optimizer(model, hyp, opt.optimizer)
    lf = one_cycle(1, hyp['lrf'], opt.epochs) if opt.cos_lr else (lambda x: (1 - x / opt.epochs) * (1.0 - hyp['lrf']) + hyp['lrf'])
    scheduler = lr_scheduler.LambdaLR(optimizer, lr_lambda=lf)
    ema = ModelEMA(model) if RANK in (-1, 0) else None
    if ckpt is not None and isinstance(ckpt, dict) and opt.resume:
        if ckpt.get('optimizer') is not None:
            optimizer.load_state_dict(ckpt['optimizer'])
            best_fitness = ckpt.get('best_fitness', 0.0)
        if ema and ckpt.get('ema'):
            ema.ema.load_state_dict(ckpt['ema'])
            ema.updates = ckpt.get('updates', 0)
        start_epoch = ckpt.get('epoch', -1) + 1

    # per-header loss gains scaled to layers / classes / image size (train.py:335-345)
    for header in model.headers.values():
        nl = header.nl
        h = header.det_loss.hyp
        h['box'] *= 3 / nl
        h['cls'] *= header.nc / 80 * 3 / nl
        h['obj'] *= (imgsz / 640) ** 2 * 3 / nl
    model.half()                                                       # bf16 operands, fp32 master weights
    net = DataParallel(model) if WORLD_SIZE > 1 else model

    loader = SyntheticTiles(batch_size, imgsz, opt.nc, opt.steps_per_epoch, rank=max(RANK, 0), seed=opt.seed, device=device)
    val_loader = SyntheticTiles(batch_size, imgsz, opt.nc, opt.val_batches, rank=0, seed=opt.seed + 99, device=device)
    nb = len(loader)
    nw = max(round(hyp['warmup_epochs'] * nb), 100)
    last_opt_step = -1
    scheduler.last_epoch = start_epoch - 1
    stopper = EarlyStopping(patience=opt.patience)
    LOGGER.info(f'Image sizes {imgsz}, batch {opt.batch_size} ({batch_size}/rank x {WORLD_SIZE}), accumulate {accumulate}, '
                f'{nb} iterations/epoch, {opt.epochs} epochs, saving to {save_dir}')
    t0 = time.time()
    for epoch in range(start_epoch, opt.epochs):
        model.train()
        loader.set_epoch(epoch)
        mloss = {}
        optimizer.zero_grad(set_to_none=True)
        for i, (imgs, targets) in enumerate(loader):
            ni = i + nb * epoch
            imgs = torch.stack(list(imgs)).to(device, non_blocking=True)
            if ni <= nw:                                               # warm-up (train.py:436-444)
                xi = [0, nw]
                accumulate = max(1, round(float(torch.tensor(ni / nw * (nbs / opt.batch_size - 1) + 1).clamp(min=1)))) if nbs > opt.batch_size else 1
                for j, g in enumerate(optimizer.param_groups):
                    lo = hyp['warmup_bias_lr'] if j == 2 else 0.0
                    g['lr'] = lo + (g['initial_lr'] * lf(epoch) - lo) * ni / nw
                    if 'momentum' in g:
                        g['momentum'] = hyp['warmup_momentum'] + (hyp['momentum'] - hyp['warmup_momentum']) * ni / nw
            losses, _ = net(imgs, targets, compute_masks=False)
            loss = sum(v['det_loss'] + v['mask_loss'] for v in losses.values())
            loss.backward()
            if ni - last_opt_step >= accumulate:
                optimizer.step()
                optimizer.zero_grad(set_to_none=True)
                if ema:
                    ema.update(model)
                last_opt_step = ni
            if RANK in (-1, 0) and (i % opt.log_every == 0 or i == nb - 1):
                for task_id, v in losses.items():
                    for k, item in v['loss_items'].items():
                        mloss[f'{task_id}/{k}'] = float(item)
                LOGGER.info(f'epoch {epoch}/{opt.epochs - 1} it {i}/{nb - 1} loss {float(loss):.4f} ' +
                            ' '.join(f'{k} {v:.4f}' for k, v in mloss.items()))
        scheduler.step()
        if RANK in (-1, 0):
            final = epoch + 1 == opt.epochs
            fitness = 0.0
            if not opt.noval or final:
                fitness, _, speeds = val.run(ema.ema, val_loader, half=True)
                ema.ema.float()
            best_fitness = max(best_fitness, fitness)
            if not opt.nosave or final:
                eng = model.__dict__.pop('_hdy_engine', None)          # device plans are not checkpoint state
                ckpt = {'epoch': epoch, 'best_fitness': best_fitness, 'model': deepcopy(de_parallel(model)).state_dict(),
                        'ema': deepcopy(ema.ema).state_dict(), 'updates': ema.updates, 'optimizer': optimizer.state_dict(),
                        'date': time.strftime('%Y-%m-%d %H:%M:%S')}
                if eng is not None:
                    object.__setattr__(model, '_hdy_engine', eng)
                torch.save(ckpt, w / 'last.pt')
                if best_fitness == fitness:
                    torch.save(ckpt, w / 'best.pt')
            if stopper(epoch=epoch, fitness=fitness):
                break
    if RANK in (-1, 0):
        LOGGER.info(f'{opt.epochs - start_epoch} epochs completed in {(time.time() - t0) / 3600:.3f} hours.')
    return best_fitness


def argument_parser():
    p = argparse.ArgumentParser()
    p.add_argument('--weights', default='')
    p.add_argument('--variant', default='s', help='n | s | m | l (stock depth/width multiples in the metayolo schema)')
    p.add_argument('--nc', type=int, default=8)
    p.add_argument('--hyp', default='', help='optional hyp yaml; defaults to the YOLOv5 scratch values')
    p.add_argument('--epochs', type=int, default=2)
    p.add_argument('--steps-per-epoch', type=int, default=20)
    p.add_argument('--val-batches', type=int, default=2)
    p.add_argument('--batch-size', type=int, default=64, help='total batch size for all GPUs')
    p.add_argument('--imgsz', '--img', '--img-size', type=int, default=640)
    p.add_argument('--resume', action='store_true')
    p.add_argument('--nosave', action='store_true')
    p.add_argument('--noval', action='store_true')
    p.add_argument('--device', default='')
    p.add_argument('--optimizer', choices=['SGD', 'Adam', 'AdamW'], default='SGD')
    p.add_argument('--cos-lr', action='store_true')
    p.add_argument('--patience', type=int, default=100)
    p.add_argument('--seed', type=int, default=0)
    p.add_argument('--log-every', type=int, default=10)
    p.add_argument('--project', default=os.path.join(ROOT, 'runs', 'train'))
    p.add_argument('--name', default='exp')
    p.add_argument('--exist-ok', action='store_true')
    return p


def main(opt):
    hyp = synth.make_hyp()
    if opt.hyp:
        import yaml
        with open(opt.hyp) as f:
            hyp.update(yaml.safe_load(f))
    opt.save_dir = str(increment_path(Path(opt.project) / opt.name, exist_ok=opt.exist_ok or RANK not in (-1, 0)))
    device = select_device(opt.device)
    if LOCAL_RANK != -1:
        assert opt.batch_size % WORLD_SIZE == 0, '--batch-size must be multiple of WORLD_SIZE'
        torch.cuda.set_device(LOCAL_RANK % torch.cuda.device_count())
        device = torch.device('cuda', LOCAL_RANK % torch.cuda.device_count())
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = os.environ.get('HDY_DIST_BACKEND', 'nccl')
        dist.init_process_group(backend=backend)
    try:
        return train(hyp, opt, device)
    finally:
        if WORLD_SIZE > 1 and dist.is_initialized():
            dist.destroy_process_group()


if __name__ == '__main__':
    main(argument_parser().parse_args())
