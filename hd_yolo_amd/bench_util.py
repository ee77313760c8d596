"""Measurement helpers shared by bench.py and scripts/: algorithmic work of a plan's launch records, per-launch timing, and the
C4 inference measurement (BASELINE.json configs[3])."""
import torch

from . import ops, synth


def describe(rec):
    """(label, FLOPs, read-once/write-once bytes) of a launch record (bf16 operands); (label, 0, 0) for records without a model."""
    name, a = rec[0], rec[1]
    if name == 'hdy_conv_fwd':
        N, H, W, C, K, R, S_, st, pad = a[11:20]
        Ho, Wo = ops.out_dim(H, R, st, pad), ops.out_dim(W, S_, st, pad)
        return f'fwd  {C:4d}->{K:4d} k{R} s{st} @{H}x{W}', 2.0 * N * Ho * Wo * K * C * R * S_, 2.0 * (N * H * W * C + N * Ho * Wo * K)
    if name in ('hdy_conv_dgrad', 'hdy_conv_dgrad_stats'):
        N, H, W, C, K, R, S_, st, pad = a[5:14]
        Ho, Wo = ops.out_dim(H, R, st, pad), ops.out_dim(W, S_, st, pad)
        return f'dgrd {C:4d}<-{K:4d} k{R} s{st} @{H}x{W}', 2.0 * N * Ho * Wo * K * C * R * S_, 2.0 * (N * H * W * C + N * Ho * Wo * K)
    if name == 'hdy_conv_wgrad':
        N, H, W, C, K, R, S_, st, pad = a[4:13]
        Ho, Wo = ops.out_dim(H, R, st, pad), ops.out_dim(W, S_, st, pad)
        return f'wgrd {C:4d}x{K:4d} k{R} s{st} @{H}x{W}', 2.0 * N * Ho * Wo * K * C * R * S_, 2.0 * (N * H * W * C + N * Ho * Wo * K)
    if name == 'hdy_conv_wgrad_stem_fused':
        N, H, W, K = a[11:15]
        Ho, Wo = ops.out_dim(H, 6, 2, 2), ops.out_dim(W, 6, 2, 2)
        # reads the padded image and BOTH dz and y of the unit (the BatchNorm backward is applied on the way in)
        return f'wgrd    3x{K:4d} k6 s2 @{H}x{W} +bn', 2.0 * N * Ho * Wo * K * 3 * 36, 2.0 * (N * H * W * 3 + 2 * N * Ho * Wo * K)
    if name == 'hdy_bn_act_fwd':
        M, K = a[8], a[9]
        return f'bnfw K={K} M={M}', 0.0, 2.0 * M * K * (3 if a[4] else 2)
    if name == 'hdy_bn_act_bwd':
        M, K = a[13], a[14]
        if a[8] is None:
            return f'bnst K={K} M={M}', 0.0, 2.0 * M * K * 2
        return f'bnbw K={K} M={M}', 0.0, 2.0 * M * K * 5
    if name in ('hdy_conv1x1_bwd_fused', 'hdy_conv1x1_bwd_fused_stats'):
        M, C, K = a[24], a[25], a[26]
        return f'f1x1 {C:4d}<>{K:4d} M={M}' + (' acc' if a[18] else ''), 4.0 * M * K * C, 2.0 * M * (2 * K + (3 if a[18] else 2) * C)
    if name == 'hdy_bn_act_bwd_apply':
        M, K = a[15], a[16]
        return f'bnap K={K} M={M}', 0.0, 2.0 * M * K * 3
    if name == 'hdy_bn_act_fwd_pair':
        M, K = a[9], a[10]
        return f'bnfw K={K} M={M} (pair)', 0.0, 2.0 * M * K * 2
    if name == 'hdy_bn_act_bwd_pair':
        M, K = a[18], a[19]
        if a[11] is None:
            return f'bnst K={K} M={M} (pair)', 0.0, 2.0 * M * K * 2
        return f'bnbw K={K} M={M} (pair)', 0.0, 2.0 * M * K * 5
    return name[4:], 0.0, 0.0


def flat_records(recs):
    out = []
    for rec in recs or []:
        if rec[0] == '@fork':
            out.extend(flat_records(rec[2]))
        elif rec[0][0] != '@':
            out.append(rec)
    return out


def plan_work(plan):
    """(FLOPs, bytes) of one forward + backward replay of a training plan: the sums of describe() over its conv / BatchNorm launches."""
    fl = by = 0.0
    for rec in flat_records(plan.fwd) + flat_records(plan.bwd):
        _, f, b = describe(rec)
        fl, by = fl + f, by + b
    return fl, by


def time_record(rec, reps=10, blocks=5, warm=3, both=False):
    """microseconds of one launch record replayed back to back (HIP events on the launch stream): `warm` untimed launches, then `blocks`
    blocks of `reps` launches each between two events; the result is the FASTEST block's average.  (Round 4 timed one block of 8 right after
    unrelated kernels: +-25 % from row to row of the same layer.  A block's average can only be inflated — by clocks still ramping, by
    the tail of whatever ran before — so the minimum over blocks is the reproducible figure; scripts/layer_bench.py uses this function too.)
    both=True returns (fastest, median) — the consumers that publish a figure report both (the minimum alone reads optimistic)."""
    for _ in range(warm):
        ops.run([rec])
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(blocks + 1)]
    evs[0].record()
    for b in range(blocks):
        for _ in range(reps):
            ops.run([rec])
        evs[b + 1].record()
    torch.cuda.synchronize()
    ts = sorted(evs[b].elapsed_time(evs[b + 1]) / reps * 1e3 for b in range(blocks))
    return (ts[0], ts[len(ts) // 2]) if both else ts[0]


def conv3x3_table(plan, peak_tflops=2500.0, reps=10, peak_gbs=8000.0):
    """Every 3x3 convolution launch (forward, and data gradient) of a plan timed alone.  Per layer: time, TFLOP/s, fraction of the
    MFMA peak, and — because the stride-2 / narrow layers cannot reach 30 % of the MFMA peak even at the HBM roofline — the roofline
    that bounds it (max(flop / MFMA peak, read-once/write-once bytes / HBM peak)) and the fraction of THAT bound."""
    rows, seen = [], set()
    for rec in flat_records(plan.fwd) + flat_records(plan.bwd):
        if rec[0] not in ('hdy_conv_fwd', 'hdy_conv_dgrad', 'hdy_conv_dgrad_stats'):
            continue
        label, fl, by = describe(rec)
        if ' k3 ' not in label or label in seen:
            continue
        seen.add(label)
        us, us_med = time_record(rec, reps, both=True)
        t_mfma, t_hbm = fl / peak_tflops / 1e6, by / peak_gbs / 1e3          # microseconds at either peak
        rows.append({'layer': label, 'us': round(us, 1), 'us_median': round(us_med, 1), 'tflops': round(fl / us / 1e6, 1), 'frac': round(fl / us / 1e6 / peak_tflops, 4),
                     'bound': 'mfma' if t_mfma >= t_hbm else 'hbm', 'frac_of_bound': round(max(t_mfma, t_hbm) / us, 4)})
    return rows


def hbm_kernel_roofline(plan, peak_gbs=8000.0, reps=10):
    """The step's dominant HBM-bound kernel — the BatchNorm-backward pass over the largest activation — timed alone:
    achieved = algorithmic bytes (read dz, read y [, write dy]) / time."""
    best = None
    for rec in flat_records(plan.bwd):
        if rec[0] not in ('hdy_bn_act_bwd', 'hdy_bn_act_bwd_pair'):
            continue
        label, _, by = describe(rec)
        if best is None or by > best[2]:
            best = (rec, label, by)
    if best is None:
        return None
    us, us_med = time_record(best[0], reps, both=True)
    out = {'bound': 'hbm', 'kernel': best[0][0][4:] + ' ' + best[1], 'achieved': round(best[2] / us / 1e3, 1), 'peak': peak_gbs, 'unit': 'GB/s',
           'frac': round(best[2] / us / 1e3 / peak_gbs, 4), 'us_per_launch': round(us, 1), 'us_per_launch_median': round(us_med, 1), 'algorithmic_mb_per_launch': round(best[2] / 1e6, 1)}
    # the same family over ALL its in-plan launches (bn_act_bwd_reduce + finalize + apply per call: the kernel family with the most time in the
    # step, profiles/r03_step_kernel_stats.txt), each replayed alone: sum of algorithmic bytes / sum of times
    tot_b, tot_us, n = 0.0, 0.0, 0
    for rec in flat_records(plan.bwd):
        if rec[0] in ('hdy_bn_act_bwd', 'hdy_bn_act_bwd_pair'):
            tot_b += describe(rec)[2]
            tot_us += time_record(rec, 5, 3)
            n += 1
    out['all_launches'] = {'n': n, 'us_total': round(tot_us, 1), 'algorithmic_gb': round(tot_b / 1e9, 3), 'achieved': round(tot_b / tot_us / 1e3, 1),
                           'frac': round(tot_b / tot_us / 1e3 / peak_gbs, 4)}
    return out


def timed(fn, n):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def infer_benchmark(variant='l', B=128, S=1024, iters=3, device=None, nc=8, survivors=1024, cpu_nms=None):
    """BASELINE.json configs[3] (C4): yolov5l, batch 128, 1024x1024, bf16 inference = eval launch list + decode + NMS + outputs."""
    from metayolo.models.yolo import Model
    dev = device or torch.device('cuda', 0)
    m = Model(synth.make_cfg(variant, nc), synth.make_hyp())
    m.load_state_dict(synth.synth_state_dict(synth.shapes_of(m), seed=0), strict=False)
    m = m.to(dev).eval().half()
    x = synth.synth_images(B, S, seed=0).to(dev)
    det_spread = synth.calibrate_det_logits(m, x[:2].contiguous())       # random-init logits of +-1e5 -> unit spread (see its docstring)
    with torch.no_grad():
        ms_all = timed(lambda: m(x), iters)
        plan = [pl for pl in m._eng().plans.values() if pl.det_views()[0].shape[0] == B][-1]
        ms_net = timed(lambda: plan.run_forward(x), iters)
        head = m.headers['det']
        dets = plan.det_views()
        ms_dec = timed(lambda: head.decode_all(dets), iters)
        preds = head.decode_all(dets)
        # Random-init logits leave next to nothing above the reference's default confidence (0.15): the NMS would be timed on empty tiles.  Histology tiles
        # hold 10^3 - 10^4 nuclei, so the threshold is set where tile 0 keeps ~1024 candidates: the filter, sort and greedy pass all have work.
        conf_used = synth.dense_conf_thres(preds[0], survivors)
        head.nms_params = dict(head.nms_params, conf_thres=conf_used)
        ms_all = timed(lambda: m(x), iters)                                  # end to end at that threshold
        ms_out = timed(lambda: head.compute_outputs(preds), iters)
        # a timing of an empty or broken result is not a measurement: every tile must keep detections and every box must be finite
        outs = head.compute_outputs(preds)
        n_keep = [len(o['boxes']) for o in outs]
        assert len(outs) == B and min(n_keep) > 0, f'inference benchmark: tiles without detections (min {min(n_keep)} of {B} tiles)'
        assert all(bool(torch.isfinite(o['boxes']).all()) for o in outs), 'inference benchmark: non-finite boxes'
        p = head.nms_params
        ms_nms = timed(lambda: ops.nms_batched(preds, head.nc, p['conf_thres'], p['iou_thres'], int(p['max_det'])), iters)
        n_surv = int(((preds[:, :, 4] > conf_used) & (preds[:, :, 2] >= 2) & (preds[:, :, 3] >= 2)).sum()) / B
    ncand = preds.shape[1]
    dec_bytes = B * ncand * (head.no + head.no + 1) * 4        # logits read + rows written (108 B per candidate at nc = 8)
    gf = {'n': 4.13, 's': 15.81, 'm': 47.94, 'l': 107.76}[variant[0]] * (S / 640) ** 2
    out = {'workload': f'yolov5{variant} {nc}-class, batch {B}, {S}x{S}, bf16 inference: network + decode + NMS + outputs', 'tiles_per_s': round(B / ms_all * 1e3, 1),
           'ms_per_batch': round(ms_all, 3), 'ms_network': round(ms_net, 3), 'network_tflops': round(gf * B / ms_net, 1),
           'decode_us_per_tile': round(ms_dec / B * 1e3, 2), 'decode_hbm_frac': round(dec_bytes / ms_dec / 1e6 / 8000, 3),
           'nms_kernel_us_per_tile': round(ms_nms / B * 1e3, 2), 'nms_plus_outputs_us_per_tile': round(ms_out / B * 1e3, 2), 'candidates_per_tile': ncand,
           'conf_thres': round(conf_used, 5), 'survivors_per_tile': round(n_surv, 1), 'detections_per_tile': [min(n_keep), round(sum(n_keep) / B, 1), max(n_keep)]}
    if cpu_nms is not None:
        # the caller's CPU leg (bench.py: the C restatement of torchvision's greedy NMS, test infrastructure) on a bounded sample of the SAME decoded tiles
        try:
            out['cpu_nms'] = cpu_nms(preds[:16].float().cpu().numpy(), head.nc, float(p['conf_thres']), float(p['iou_thres']), int(p['max_det']))
        except Exception as e:          # noqa: BLE001 — no gcc on the box, a failed build: the GPU measurement above must survive its CPU side leg
            out['cpu_nms'] = {'error': repr(e)[:200]}
    del m, x, preds, dets
    torch.cuda.empty_cache()
    return out
