#!/usr/bin/env python3
"""Inference caller on the MI355X path (reference: evaluation.py:27-66 build_model, :69-80 attempt_load_model,
:83-150 inference_on_loader_yolov5, and the whole-slide helpers Detect.merge_outputs / rescale_outputs, yolo_head.py:450-471).

    build_model(path | state | Model, ref_model=None, half=True, extra_configs={}) -> (model, deployed)
    attempt_load_model(path | [paths], ...)                                        -> single pair, or (Ensemble, Ensemble)
    inference_on_loader_yolov5(deployed, loader, device, input_size=640, compute_masks=False) -> (results, seconds per image)
    inference_on_slide(deployed, slide (3, H, W), tile=640, overlap=64, ...)       -> one merged {'boxes','scores','labels'} per task

What differs from the reference, on purpose: `torch.jit.script(Deploy(model))` has no counterpart — the eval launch list of the
wrapped Model is the deployed artefact (yolo.Deploy) —, checkpoints may hold state_dicts instead of pickled modules
(engines/general.checkpoint_state reads both), and display / plotting / pandas evaluation tables are CPU-side reporting outside the
hot path.  Timing brackets exactly what the reference brackets (:98-105: resize + model call), with a device
synchronisation on both sides because HIP launches are asynchronous.

    python evaluation.py --variant s --nc 8 --imgsz 640 --batch-size 32 --batches 4 [--weights w.pt] [--slide 2048]
"""
import argparse
import os
import sys
import time
from collections import OrderedDict

os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')               # before the HIP runtime loads: see hd_yolo_amd/__init__.py
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault('YOLOv5_VERBOSE', 'false')

from metayolo.engines.general import checkpoint_state, intersect_dicts, manipulate_header_label_order  # noqa: E402
from metayolo.models.utils_general import nms, scale_coords  # noqa: E402
from metayolo.models.yolo import Deploy, Ensemble, Model  # noqa: E402
from val_nuclei import flatten_onehot_objects  # noqa: E402


@torch.no_grad()
def build_model(model_path, ref_model=None, half=True, extra_configs={}):
    """Checkpoint (path, loaded checkpoint or state_dict) -> (Model in eval mode, Deploy wrapper).  `ref_model` supplies cfg / hyp
    when the checkpoint is a bare state_dict (reference :30-36); EMA weights are preferred (:34); anchors come from the cfg (:37)."""
    ckpt = torch.load(model_path, map_location='cpu', weights_only=False) if isinstance(model_path, (str, os.PathLike)) else model_path
    if isinstance(ckpt, Model):
        ref_model, csd = ref_model or ckpt, ckpt.state_dict()
    else:
        holder = None
        if isinstance(ckpt, dict) and 'model' in ckpt:
            holder = ckpt['ema'] if ckpt.get('ema') is not None else ckpt['model']
        if isinstance(holder, torch.nn.Module):
            ref_model = ref_model or holder
        assert ref_model is not None, 'model cannot be None if only state_dict is given.'
        csd = checkpoint_state(ckpt, prefer_ema=True)
    csd = OrderedDict((k, v) for k, v in csd.items() if 'anchor' not in k)
    model = Model(ref_model.cfg, ref_model.hyp)
    model.load_state_dict(intersect_dicts(csd, model.state_dict()), strict=False)
    for key, cfgs in extra_configs.get('headers', {}).items():         # {'headers': {'det': {'label_map': [...], 'nms_params': {...}}}}
        if 'label_map' in cfgs:
            model.headers[key] = manipulate_header_label_order(model.headers[key], cfgs['label_map'])
        if 'nms_params' in cfgs:
            model.headers[key].nms_params = model.headers[key].get_nms_params(cfgs['nms_params'])
    model.eval()
    if half:
        model.half()
    return model, Deploy(model)


@torch.no_grad()
def attempt_load_model(weights_path, ref_model=None, half=True, extra_configs={}):
    if not isinstance(weights_path, (list, tuple)):
        return build_model(weights_path, ref_model=ref_model, half=half, extra_configs=extra_configs)
    pairs = [build_model(p, ref_model=ref_model, half=half, extra_configs=extra_configs) for p in weights_path]
    return Ensemble([m for m, _ in pairs]), Ensemble([d for _, d in pairs])


@torch.no_grad()
def inference_on_loader_yolov5(model, data_loader, device, input_size=640, compute_masks=False, **kwargs):
    """Timed loop of the reference (:83-150): stack, bilinear resize to `input_size`, model call, boxes back to the original frame with
    scale_coords(...).round(), multi-hot labels flattened, everything to the host.  Returns (list of per-image dicts, s / image)."""
    model.eval()
    model.to(device)
    results, total_time, n_images = [], 0.0, 0
    for images, _targets in data_loader:
        images = torch.stack(list(images)).to(device, non_blocking=True)
        ori_size = tuple(images.shape[-2:])
        torch.cuda.synchronize(device)
        st = time.time()
        size = (input_size, input_size) if isinstance(input_size, int) else tuple(input_size)
        inputs = images if size == ori_size else torch.nn.functional.interpolate(images.float(), size=size, mode='bilinear', align_corners=False)
        _, outputs = model(inputs, compute_masks=compute_masks)
        torch.cuda.synchronize(device)
        total_time += time.time() - st
        n_images += len(images)
        for output in outputs:
            for task_id in output:
                o = output[task_id]
                o['boxes'] = scale_coords(size, o['boxes'], ori_size).round()
                if o['labels'].dim() == 2:
                    o = flatten_onehot_objects(o)
                output[task_id] = {k: v.detach().cpu() for k, v in o.items()}
            results.append(output)
    return results, total_time / max(n_images, 1)


def slide_rois(height, width, tile, overlap):
    """Top-left corners (x0, y0) of `tile`-sized windows covering the slide with at least `overlap` pixels shared between neighbours;
    the last window of a row / column is shifted back inside the slide."""
    def starts(n):
        if n <= tile:
            return [0]
        step = tile - overlap
        s = list(range(0, n - tile, step)) + [n - tile]
        return sorted(set(s))
    return [(x0, y0) for y0 in starts(height) for x0 in starts(width)]


@torch.no_grad()
def inference_on_slide(model, slide, tile=640, overlap=64, batch_size=32, scale=1.0, iou_thres=None, compute_masks=False):
    """Whole-slide detection as the reference's ROI protocol composes it: tiles of one amplification are run in batches, each tile's
    detections carry their 'roi' offset, `Detect.merge_outputs` shifts and concatenates them (yolo_head.py:450-462), overlapping
    windows are de-duplicated by one class-agnostic NMS on the MI355X kernel (as Ensemble.merge does, yolo.py:189-199), and
    `Detect.rescale_outputs` maps the boxes to another amplification (:464-471).  slide: (3, H, W) float in 0..1 on the GPU."""
    assert slide.dim() == 3 and slide.is_cuda
    inner = model._model if isinstance(model, Deploy) else model
    _, H, W = slide.shape
    rois = slide_rois(H, W, tile, overlap)
    per_task = {}
    for i in range(0, len(rois), batch_size):
        chunk = rois[i:i + batch_size]
        x = slide.new_zeros((len(chunk), 3, tile, tile))
        for j, (x0, y0) in enumerate(chunk):
            patch = slide[:, y0:y0 + tile, x0:x0 + tile]
            x[j, :, :patch.shape[1], :patch.shape[2]] = patch
        _, outputs = model(x, compute_masks=compute_masks)
        for (x0, y0), out in zip(chunk, outputs):
            for task_id, o in out.items():
                per_task.setdefault(task_id, []).append(dict(o, roi=(float(x0), float(y0))))
    merged = {}
    for task_id, parts in per_task.items():
        header = inner.headers[task_id]
        r = header.merge_outputs(parts)
        thr = header.nms_params['iou_thres'] if iou_thres is None else iou_thres
        if len(r['boxes']) and overlap > 0:
            keep = nms(r['boxes'], r['scores'], thr)
            r = {k: v[keep] for k, v in r.items()}
        r['boxes'][:, [0, 2]] = r['boxes'][:, [0, 2]].clamp(0, W)
        r['boxes'][:, [1, 3]] = r['boxes'][:, [1, 3]].clamp(0, H)
        merged[task_id] = header.rescale_outputs(r, scale)
    return merged


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--weights', nargs='*', default=[], help='checkpoint(s); several = Ensemble; none = synthetic weights')
    ap.add_argument('--variant', default='s')
    ap.add_argument('--nc', type=int, default=8)
    ap.add_argument('--imgsz', type=int, default=640, help='network input size')
    ap.add_argument('--tile', type=int, default=0, help='original tile size (resized to --imgsz); default = --imgsz')
    ap.add_argument('--batch-size', type=int, default=32)
    ap.add_argument('--batches', type=int, default=4)
    ap.add_argument('--slide', type=int, default=0, help='also run one synthetic SxS slide through inference_on_slide')
    ap.add_argument('--no-half', action='store_true')
    ap.add_argument('--device', default='')
    opt = ap.parse_args()
    from hd_yolo_amd import synth
    from metayolo.datasets import SyntheticTiles
    from metayolo.engines.torch_utils import select_device
    device = select_device(opt.device)
    ref = Model(synth.make_cfg(opt.variant, opt.nc), synth.make_hyp())
    if opt.weights:
        w = opt.weights if len(opt.weights) > 1 else opt.weights[0]
        model, deployed = attempt_load_model(w, ref_model=ref, half=not opt.no_half)
    else:
        ref.load_state_dict(synth.synth_state_dict(synth.shapes_of(ref), seed=0), strict=False)
        model, deployed = build_model(ref, half=not opt.no_half)
    loader = SyntheticTiles(opt.batch_size, opt.tile or opt.imgsz, opt.nc, opt.batches, seed=2024)
    inference_on_loader_yolov5(deployed, SyntheticTiles(opt.batch_size, opt.tile or opt.imgsz, opt.nc, 1, seed=1), device, input_size=opt.imgsz)   # warm-up: plans
    results, spi = inference_on_loader_yolov5(deployed, loader, device, input_size=opt.imgsz)
    n = sum(len(next(iter(r.values()))['boxes']) for r in results)
    print(f'{len(results)} tiles, {n} detections, {spi * 1e3:.3f} ms / tile ({1.0 / spi:.0f} tiles/s incl. host transfer of the results)')
    if opt.slide:
        slide = synth.synth_images(1, opt.slide, seed=5)[0].to(device)
        torch.cuda.synchronize()
        t0 = time.time()
        out = inference_on_slide(deployed.to(device), slide, tile=opt.imgsz, batch_size=opt.batch_size)
        torch.cuda.synchronize()
        print(f'slide {opt.slide}x{opt.slide}: ' + ', '.join(f'{k}: {len(v["boxes"])} detections' for k, v in out.items()) + f' in {(time.time() - t0) * 1e3:.1f} ms')


if __name__ == '__main__':
    main()
