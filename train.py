#!/usr/bin/env python3
"""Training entry point with the reference's structure (reference: train.py:87-588 train(), :591-641 argument_parser,
:644-690 main) on the MI355X path, for synthetic tiles.  Kept from the reference because they change results:
nominal batch 64 / accumulate / scaled weight decay (:208-210), three parameter groups (:213-226), SGD-Nesterov | Adam |
AdamW (:229-233), linear or one-cycle LambdaLR (:242-246), warm-up of lr and momentum (:354, :436-444), EMA (:249),
per-header loss-gain rescale (:335-345), one process per GPU with env:// rendezvous (:67-69, :683).
Replaced: DistributedDataParallel -> hd_yolo_amd.parallel.DataParallel (flat-bucket RCCL sum all-reduce; the loss is NOT
multiplied by WORLD_SIZE, cf. :467); amp.autocast/GradScaler -> bf16 operands with fp32 masters (no loss scaling needed);
datasets/loggers/plots/evolve/W&B -> out of scope.

Entry-point surface kept from the reference's argument_parser (:594-639): --weights --cfg --hyp --epochs --batch-size --imgsz
--resume --nosave --noval --device --optimizer --sync-bn --project --name --exist-ok --cos-lr --label-smoothing --patience
--freeze --save-period --masks --restart.  --cfg takes a metayolo-schema yaml (backbone / fpn / headers, as the hub files
metayolo/hub/yolov5{m6,l6}-multihead.yaml, yolov5l6-mask.yaml) or is replaced by --variant n|s|m|l|n6|m6|l6 for the stock graphs.
Checkpoints: state_dicts by default ('model', 'ema'); `--weights` also reads the reference's pickled-module checkpoints
(`ckpt['model'].float().state_dict()`, EMA preferred with --restart) and stock YOLOv5 key layouts (convert_yolo_weights).

    python train.py --variant s --nc 8 --batch-size 64 --imgsz 640 --epochs 2 --steps-per-epoch 20
    python train.py --cfg my_model.yaml --hyp my_hyp.yaml --freeze backbone --masks --save-period 5
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train.py --batch-size 512 ...
"""
import argparse
import json
import math
import os
import random
import sys
import time
from copy import deepcopy
from pathlib import Path

os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')               # before the HIP runtime loads: see hd_yolo_amd/__init__.py
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
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
from metayolo.engines.general import checkpoint_state, convert_yolo_weights, increment_path, init_seeds, intersect_dicts, one_cycle  # noqa: E402
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
        from hd_yolo_amd.optim import SGD as FusedSGD          # torch.optim.SGD's update and state layout, one launch for all tensors
        opt = FusedSGD(g_bn, lr=hyp['lr0'], momentum=hyp['momentum'], nesterov=True)
    opt.add_param_group({'params': g_w, 'weight_decay': hyp['weight_decay']})
    opt.add_param_group({'params': g_b})
    return opt


def load_pretrained(model, path, resume=False, restart=False, has_cfg=True):
    ckpt = torch.load(path, map_location='cpu', weights_only=False)
    csd = checkpoint_state(ckpt, prefer_ema=restart)
    if any(k.startswith('model.') for k in csd):                    # stock YOLOv5 key layout (train.py:170-173)
        csd = convert_yolo_weights(model, csd)
    exclude = ['anchor'] if has_cfg and not resume else []          # train.py:169
    csd = intersect_dicts(csd, model.state_dict(), exclude=exclude)
    model.load_state_dict(csd, strict=False)
    LOGGER.info(f'Transferred {len(csd)}/{len(model.state_dict())} items from {path}')
    return ckpt if isinstance(ckpt, dict) and 'model' in ckpt else None


def train(hyp, opt, device):
    save_dir = Path(opt.save_dir)
    w = save_dir / 'weights'
    if RANK in (-1, 0):
        w.mkdir(parents=True, exist_ok=True)
        with open(save_dir / 'opt.json', 'w') as f:
            json.dump(vars(opt), f, indent=1, default=str)
        import yaml
        with open(save_dir / 'hyp.yaml', 'w') as f:
            yaml.safe_dump(hyp, f, sort_keys=False)
    init_seeds(1 + RANK)
    if opt.cfg:
        cfg = opt.cfg                                                  # metayolo-schema yaml: Model reads it (metayolo.load_cfg)
    else:
        cfg = synth.make_cfg(opt.variant, opt.nc)
        if opt.masks:                                                  # stock graph + the mask branch (one shared mask class, yolov5l6-mask.yaml:64)
            cfg['headers'][0][3][3] = 1
    model = Model(cfg, hyp, ch=3, anchors=hyp.get('anchors'))
    start_epoch, best_fitness = 0, 0.0
    ckpt = None
    if opt.weights:
        ckpt = load_pretrained(model, opt.weights, resume=bool(opt.resume), restart=opt.restart, has_cfg=True)
        if opt.restart:
            ckpt = None
    model = model.to(device)
    model = model.freeze(opt.freeze)                                   # train.py:182, `--freeze backbone neck.0 headers.det.m ...`
    task0 = next(iter(model.headers))
    nc = model.headers[task0].nc
    with_masks = opt.masks and any(getattr(h, 'nc_masks', 0) > 0 for h in model.headers.values())
    if opt.masks and not with_masks:
        raise SystemExit('--masks: the model config has no mask branch (Detect args[3] must be >= 0)')
    if opt.sync_bn and WORLD_SIZE == 1:
        LOGGER.info('--sync-bn has no effect with one process (the reference converts the model only in DDP mode, train.py:281-283)')
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
        if ema and ckpt.get('ema') is not None:
            ema.ema.load_state_dict(intersect_dicts(checkpoint_state({'model': ckpt['ema']}), ema.ema.state_dict()), strict=False)
            ema.updates = ckpt.get('updates', 0)
        start_epoch = ckpt.get('epoch', 0)                             # checkpoints store epoch + 1, as the reference (train.py:531)
        # train.py:267: a finished run's checkpoint (epoch -1 once its optimizer state is stripped) has nothing to resume
        if start_epoch is None or start_epoch < 0 or ckpt.get('optimizer') is None:
            raise SystemExit(f'--resume: {opt.weights} is a finished / stripped checkpoint (epoch {start_epoch}, optimizer '
                             f'{"present" if ckpt.get("optimizer") is not None else "absent"}): nothing to resume; start a new run with --weights alone')
        if opt.epochs < start_epoch:                                   # train.py:268-270: fine-tune for `epochs` more
            LOGGER.info(f'{opt.weights} has been trained for {start_epoch} epochs. Fine-tuning for {opt.epochs} more epochs.')
            opt.epochs += start_epoch

    # per-header loss gains scaled to layers / classes / image size (train.py:335-345)
    for header in model.headers.values():
        nl = header.nl
        h = header.det_loss.hyp
        h['box'] *= 3 / nl
        h['cls'] *= header.nc / 80 * 3 / nl
        h['obj'] *= (imgsz / 640) ** 2 * 3 / nl
    model.half()                                                       # bf16 operands, fp32 master weights
    net = DataParallel(model, sync_bn=opt.sync_bn) if WORLD_SIZE > 1 else model       # --sync-bn: BatchNorm statistics over all ranks (train.py:281-283)

    mk = dict(task=task0, masks=with_masks)
    if with_masks:
        mk.update(nmin=4, nmax=24)                                     # every object carries a 28x28 mask target
    loader = SyntheticTiles(batch_size, imgsz, nc, opt.steps_per_epoch, rank=max(RANK, 0), seed=opt.seed, device=device, **mk)
    val_loader = SyntheticTiles(batch_size, imgsz, nc, opt.val_batches, rank=0, seed=opt.seed + 99, device=device, task=task0)
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
            if opt.multi_scale > 0.0:                                  # train.py:447-452: a random size within +/- multi_scale of imgsz, multiple of the stride
                sz = random.randrange(int(imgsz * (1 - opt.multi_scale)), int(imgsz * (1 + opt.multi_scale)) + gs) // gs * gs
                sf = sz / max(imgs.shape[2:])
                if sf != 1:
                    ns = [math.ceil(v * sf / gs) * gs for v in imgs.shape[2:]]
                    imgs, targets = rescale_train_batch(imgs, targets, ns)
            if ni <= nw:                                               # warm-up (train.py:436-444)
                xi = [0, nw]
                accumulate = max(1, round(float(torch.tensor(ni / nw * (nbs / opt.batch_size - 1) + 1).clamp(min=1)))) if nbs > opt.batch_size else 1
                for j, g in enumerate(optimizer.param_groups):
                    lo = hyp['warmup_bias_lr'] if j == 2 else 0.0
                    g['lr'] = lo + (g['initial_lr'] * lf(epoch) - lo) * ni / nw
                    if 'momentum' in g:
                        g['momentum'] = hyp['warmup_momentum'] + (hyp['momentum'] - hyp['warmup_momentum']) * ni / nw
            losses, _ = net(imgs, targets, compute_masks=with_masks)
            loss = sum(v['det_loss'] + (v['mask_loss'] if with_masks and 'mask_loss' in v else 0.0) for v in losses.values())   # train.py:459-462
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
        stop = False
        if RANK in (-1, 0):
            final = epoch + 1 == opt.epochs
            fitness = 0.0
            if not opt.noval or final:
                fitness, _, speeds = val.run(ema.ema, val_loader, half=True)
                ema.ema.float()
            best_fitness = max(best_fitness, fitness)
            if not opt.nosave or final:
                eng = model.__dict__.pop('_hdy_engine', None)          # device plans are not checkpoint state
                ckpt = {'epoch': epoch + 1, 'best_fitness': best_fitness, 'model': deepcopy(de_parallel(model)).state_dict(),
                        'ema': deepcopy(ema.ema).state_dict(), 'updates': ema.updates, 'optimizer': optimizer.state_dict(),
                        'date': time.strftime('%Y-%m-%d %H:%M:%S')}
                if eng is not None:
                    object.__setattr__(model, '_hdy_engine', eng)
                torch.save(ckpt, w / 'last.pt')
                if best_fitness == fitness:
                    torch.save(ckpt, w / 'best.pt')
                if opt.save_period > 0 and (epoch + 1) % opt.save_period == 0:          # train.py:544
                    torch.save(ckpt, w / f'epoch{epoch + 1}.pt')
            stop = bool(stopper(epoch=epoch, fitness=fitness))
        # The reference stops early only in single-GPU runs (train.py:550: under DDP only rank 0 would leave the loop and the others
        # would hang in the next all-reduce).  Here rank 0's decision is shared, so every rank leaves together.
        if WORLD_SIZE > 1 and dist.is_initialized():
            flag = [stop]
            dist.broadcast_object_list(flag, 0)
            stop = flag[0]
        if stop:
            break
    if RANK in (-1, 0):
        LOGGER.info(f'{opt.epochs - start_epoch} epochs completed in {(time.time() - t0) / 3600:.3f} hours.')
    return best_fitness


def argument_parser():
    p = argparse.ArgumentParser()
    p.add_argument('--weights', default='', help='initial weights: this build\'s or the reference\'s checkpoint, or a bare state_dict')
    p.add_argument('--cfg', default='', help='model.yaml path in the metayolo schema (backbone / fpn / headers); overrides --variant / --nc')
    p.add_argument('--multi-scale', type=float, default=0.0, help='vary img-size +/- multi-scale%% (train.py:610): every size gets its own cached plan')
    p.add_argument('--variant', default='s', help='n | s | m | l | n6 | m6 | l6 (stock depth/width multiples in the metayolo schema)')
    p.add_argument('--nc', type=int, default=8)
    p.add_argument('--hyp', default='', help='hyperparameters yaml (train keys + one sub-dict per header tag); defaults to the YOLOv5 scratch values')
    p.add_argument('--epochs', type=int, default=2)
    p.add_argument('--steps-per-epoch', type=int, default=20)
    p.add_argument('--val-batches', type=int, default=2)
    p.add_argument('--batch-size', type=int, default=64, help='total batch size for all GPUs')
    p.add_argument('--imgsz', '--img', '--img-size', type=int, default=640)
    p.add_argument('--resume', action='store_true', help='continue from --weights: optimizer, EMA, epoch counter')
    p.add_argument('--restart', action='store_true', help='keep only the (EMA) weights of --weights, reinitialise everything else')
    p.add_argument('--freeze', nargs='+', type=str, default=[], help='Freeze layers: `backbone`, `neck.0`, `headers.det.m`, etc')
    p.add_argument('--masks', action='store_true', help='Train mask header.')
    p.add_argument('--label-smoothing', type=float, default=0.0, help='Label smoothing epsilon')
    p.add_argument('--save-period', type=int, default=-1, help='Save checkpoint every x epochs (disabled if < 1)')
    p.add_argument('--sync-bn', action='store_true', help='use SyncBatchNorm, only available in DDP mode')
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


def rescale_train_batch(imgs, tgts, size):
    """train.py:72-80: bilinear resize of the tile batch; the boxes are normalised in training, only the recorded sizes change.
    Every distinct size gets its own static plan (engine plan cache, the oldest evicted beyond its capacity)."""
    imgs = torch.nn.functional.interpolate(imgs, size=size, mode='bilinear', align_corners=False)
    for tgt in tgts:
        tgt['size'] = torch.tensor(size, dtype=tgt['size'].dtype) if torch.is_tensor(tgt['size']) else type(tgt['size'])(size)
        for anns in tgt['anns'].values():
            for a in anns:
                a['size'] = torch.tensor(size, dtype=a['size'].dtype) if torch.is_tensor(a['size']) else type(a['size'])(size)
    return imgs, tgts


def main(opt):
    hyp = synth.make_hyp()
    if opt.hyp:
        import yaml
        with open(opt.hyp) as f:
            user = yaml.safe_load(f)
        for k, v in user.items():                                       # header sub-dicts are merged key by key
            if isinstance(v, dict) and isinstance(hyp.get(k), dict):
                hyp[k].update(v)
            else:
                hyp[k] = v
    hyp['label_smoothing'] = opt.label_smoothing                         # train.py:102
    for v in hyp.values():                                               # the per-header loss dicts are what DetLoss reads (yolov5.py:104-108)
        if isinstance(v, dict) and 'box' in v:
            v['label_smoothing'] = opt.label_smoothing
    if opt.cfg:
        # a cfg may name header tags the default hyp has no sub-dict for: they start from the 'det' defaults
        from metayolo import load_cfg
        for row in load_cfg(opt.cfg).get('headers', []):
            hyp.setdefault(row[4], deepcopy(hyp['det']))
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
