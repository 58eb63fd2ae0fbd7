#!/usr/bin/env python
"""Benchmark of the MI355X hot path: micrographs/s, denoise -> score -> NMS on 4096x4096 fp32.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload pipeline|extract|denoise]

One step = one synthetic 4096x4096 micrograph (N(0,1), seed 1000+i, already resident in HBM)
through the whole path on one GPU:
    U-Net denoise   `topaz denoise -m unet` architecture (UDenoiseNet base 11 / top 5, nf 48) with the
                    CLI-default patching -s 1024 -p 500 (16 patches, 50.4 Mpx), per-patch normalisation
    ResNet8 score   `topaz extract -m resnet8` architecture (units 64), filled, head fused
    NMS             radius 14, threshold -6  -> pick table
The pretrained blobs of both default architectures are absent upstream, so
the weights are seeded random (calibrated to realistic logit statistics) -- the arithmetic and the
shapes are those of the named configs.  Multi-GPU (torchrun): every rank processes its own K images
(weak scaling), then the pick tables are gathered to rank 0 over RCCL (the only collective).

Prints ONE JSON line (see the driver contract) including
  roofline     -- the dominant kernel class (conv_mfma, fp32 matrix cores): algorithmic FLOP / HIP-event
                  time of those launches, measured live in a separate profiled step after the timed region
  cpu_baseline -- the oracle (CPU restatement of the reference, torch-CPU) timed on a bounded sample of
                  the same workload on this host, rank 0 and N=1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3       # /opt/skills/guides/MI355X_MICROARCH.md, Peak FP32 (matrix)
F16_MFMA_PEAK_TFLOPS = 2500.0       # same table: BF16/FP16 MFMA, dense ("~2.5 PF dense", 2495 measured)
# conv_split kernels issue three f16 MFMAs per fp32-equivalent multiply-add (wh*xh + wh*xl + wl*xh), so their
# ceiling in ALGORITHMIC (fp32-equivalent) FLOP/s is a third of the f16 peak
SPLIT_PEAK_TFLOPS = F16_MFMA_PEAK_TFLOPS / 3.0


def build_models(workload: str):
    from tools import synth_weights as sw         # seeded NumPy weights; nothing under oracle/ is touched here
    from topaz_amd.denoise import Denoise
    from topaz_amd.denoising.models import DenoiseNet
    out = {}
    if workload in ('pipeline', 'denoise'):
        sd_d = sw.unet_sd(11, nf=48, base_width=11, top_width=5)
        out['denoise'] = (Denoise(DenoiseNet('unet', sd_d)), sd_d)
    if workload in ('pipeline', 'extract'):
        out['score'] = sw.hip_resnet('resnet8', 64, seed=7)      # head calibrated with the HIP path's own logits
    return out


def run_step(models, x_dev, args):
    """one micrograph through the path; returns the pick table (device tensors)"""
    from topaz_amd import runtime as rt
    img = x_dev
    if 'denoise' in models:
        img = models['denoise'][0].denoise_device(img, args.patch_size, args.patch_padding)
    if 'score' in models:
        logits = models['score'][0](img[None, None])[0, 0]
        return rt.nms(logits, args.radius, args.threshold)
    return img, None


def cpu_baseline(models, args):
    """oracle timed on a bounded sample: a SxS crop of micrograph 0 through the same stages; scaled to
    micrographs/s by the pixel counts the full workload processes (patched denoising touches 3.0x the
    image)."""
    from oracle import denoising as oden
    from oracle import nms as onms
    from oracle import scoring as oscoring
    S = args.cpu_sample
    x = np.random.RandomState(1000).randn(S, S).astype(np.float32)
    threads = torch.get_num_threads()
    per_image = 0.0
    parts = {}
    full_px = float(args.size) ** 2
    if 'denoise' in models:
        sd = models['denoise'][1]
        t0 = time.time(); den = oden.denoise('unet', sd, x, -1); t = time.time() - t0
        # pixels the full job pushes through the net with -s/-p patching
        n_px = 0
        for i in range(0, args.size, args.patch_size):
            for j in range(0, args.size, args.patch_size):
                h = min(args.size, i + args.patch_size + args.patch_padding) - max(0, i - args.patch_padding)
                w = min(args.size, j + args.patch_size + args.patch_padding) - max(0, j - args.patch_padding)
                n_px += h * w
        parts['denoise_s'] = t
        per_image += t * n_px / (S * S)
        x = den
    if 'score' in models:
        sd = models['score'][1]
        t0 = time.time(); logit = oscoring.score('resnet8', sd, x); t = time.time() - t0
        parts['score_s'] = t
        per_image += t * full_px / (S * S)
        t0 = time.time(); onms.nms2d(logit, args.radius, args.threshold); t = time.time() - t0
        parts['nms_s'] = t
        per_image += t * full_px / (S * S)
    return {'value': 1.0 / per_image, 'unit': 'micrographs/s', 'cores': threads, 'kind': 'port',
            'sample': f'{S}x{S} crop of micrograph 0 through the oracle (torch-CPU convs, C NMS), '
                      f'times {parts} scaled by processed-pixel ratio to {args.size}x{args.size}'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=4)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--workload', default='pipeline', choices=['pipeline', 'extract', 'denoise'])
    ap.add_argument('--size', type=int, default=4096)
    ap.add_argument('--patch-size', type=int, default=1024)
    ap.add_argument('--patch-padding', type=int, default=500)
    ap.add_argument('--radius', type=int, default=14)
    ap.add_argument('--threshold', type=float, default=-6.0)
    ap.add_argument('--cpu-sample', type=int, default=1024)
    ap.add_argument('--lanes', type=int, default=1, help='micrographs in flight per GPU (host threads / HIP streams)')
    ap.add_argument('--no-kernel-timing', action='store_true', help='do not record HIP events around the launches of the timed steps')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    from topaz_amd import parallel
    rank, local_rank, world = parallel.init_from_env()
    assert world == args.gpus or world == 1, f'WORLD_SIZE={world} but --gpus {args.gpus}'
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    from topaz_amd.runtime import get_context
    ctx = get_context(local_rank)

    # this rank's micrographs, resident in HBM before the timed region (global index = rank + i*world)
    n_img = args.steps + args.warmup * args.lanes
    imgs = [torch.from_numpy(np.random.RandomState(1000 + rank + i * world).randn(args.size, args.size)
                             .astype(np.float32)).to(dev) for i in range(n_img)]

    # Lanes: `--lanes L` host threads, each with its own tpz context (stream + workspace) and model copies,
    # keep L micrographs in flight on the GPU; steps are dealt round-robin, exactly K steps are timed.
    import threading
    from topaz_amd import runtime as rt
    start = threading.Barrier(args.lanes + 1)
    ready = threading.Barrier(args.lanes + 1)
    lane_models, results, errors = [None] * args.lanes, [[] for _ in range(args.lanes)], []
    lane_stats = [None] * args.lanes       # per-lane HIP-event statistics of the launches of the timed steps

    def lane_main(k):
        try:
            torch.cuda.set_device(local_rank)
            rt.set_lane(k)
            with torch.cuda.stream(torch.cuda.Stream(device=dev)):
                lane_models[k] = build_models(args.workload)
                for w in range(args.warmup):
                    run_step(lane_models[k], imgs[args.steps + k * args.warmup + w], args)
                torch.cuda.current_stream().synchronize()
                lctx = rt.get_context(local_rank)
                if not args.no_kernel_timing:
                    # roofline evidence: HIP events around the convolution launches of the TIMED steps (those of
                    # >= 20 GFLOP: ~170 of ~700 launches per step, > 95 % of the kernel time), recorded on this lane's
                    # own stream and resolved only after the timed region; costs ~0.4 % of the step
                    lctx.prof_enable(2)
                    lctx.prof_reset()
                ready.wait()
                start.wait()
                for i in range(k, args.steps, args.lanes):
                    s, c = run_step(lane_models[k], imgs[i], args)
                    if c is not None:
                        results[k].append((rank + i * world, s, c))
                torch.cuda.current_stream().synchronize()
                lane_stats[k] = (lctx, )
        except BaseException as e:          # surface worker failures in the main thread
            errors.append(e)
            for b in (ready, start):
                try:
                    b.abort()
                except Exception:
                    pass

    threads = [threading.Thread(target=lane_main, args=(k,)) for k in range(args.lanes)]
    for t in threads:
        t.start()
    try:
        ready.wait()
    except threading.BrokenBarrierError:
        pass
    if errors:
        raise errors[0]
    torch.cuda.synchronize(dev)
    parallel.barrier(dev)
    t0 = time.perf_counter()
    start.wait()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    picks = sorted((r for lane in results for r in lane), key=lambda r: r[0])
    ids, scs, cds = [p[0] for p in picks], [p[1] for p in picks], [p[2] for p in picks]
    if ids and world > 1:
        parallel.gather_pick_tables(ids, scs, cds, dev)       # the one RCCL exchange step
    torch.cuda.synchronize(dev)
    parallel.barrier(dev)
    dt = time.perf_counter() - t0
    dt = parallel.max_over_ranks(dt, dev)
    n_picks = int(sum(int(s.numel()) for s in scs)) if scs else 0
    models = lane_models[0]
    rt.set_lane(0)

    # ---- roofline: per-kernel HIP-event times of the launches of the timed steps (all lanes), per step
    merged, other, conv_ms, conv_n, conv_flops = {}, {'conv_direct_ms': 0.0, 'elementwise_ms': 0.0, 'nms_ms': 0.0}, 0.0, 0, 0.0
    if args.no_kernel_timing:
        # no events in the timed region: time one extra step on its own stream instead
        with torch.cuda.stream(torch.cuda.Stream(device=dev)):
            c0 = get_context(local_rank)
            c0.prof_enable(True)
            c0.prof_reset()
            run_step(models, imgs[-1], args)
            torch.cuda.current_stream().synchronize()
        stat_ctxs, n_prof_steps = [c0], 1
    else:
        stat_ctxs, n_prof_steps = [st[0] for st in lane_stats if st], args.steps
    for c in stat_ctxs:
        for name, ms, n, fl in c.prof_kernels():
            m0 = merged.setdefault(name, [0.0, 0, 0.0])
            m0[0] += ms; m0[1] += n; m0[2] += fl
        ms, n, fl = c.prof_get(0)
        conv_ms += ms; conv_n += n; conv_flops += fl
        for k, key in ((1, 'conv_direct_ms'), (2, 'elementwise_ms'), (3, 'nms_ms')):
            other[key] += c.prof_get(k)[0]
        c.prof_enable(False)
    kernels = sorted(((nm, v[0] / n_prof_steps, v[1] / n_prof_steps, v[2] / n_prof_steps) for nm, v in merged.items()),
                     key=lambda r: -r[1])                    # (name, ms per step, launches per step, FLOP per step)
    conv_ms, conv_n, conv_flops = conv_ms / n_prof_steps, conv_n / n_prof_steps, conv_flops / n_prof_steps
    other = {k: v / n_prof_steps for k, v in other.items()}
    dom_name, dom_ms, dom_n, dom_flops = kernels[0] if kernels else ('', 0.0, 0, 0.0)
    achieved = dom_flops / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process, so the value is the one
    # measured for exactly this kernel and shape with separate rocprofv3 --pmc passes (FETCH_SIZE + WRITE_SIZE, raw,
    # profiles/r01_split_head_pmc.txt); null for any other configuration
    traffic = None
    if dom_name.startswith('conv_split_kernel<K=5x5,D=4,MT=128') and args.size == 4096 and args.workload != 'denoise':
        traffic = 14.6e9 + 2.68e9
    is_split = dom_name.startswith('conv_split')
    peak = SPLIT_PEAK_TFLOPS if is_split else FP32_MFMA_PEAK_TFLOPS

    def klass(prefix):
        rows = [k for k in kernels if k[0].startswith(prefix)]
        ms, n, fl = sum(k[1] for k in rows), sum(k[2] for k in rows), sum(k[3] for k in rows)
        return {'kernel_ms_per_step': ms, 'launches_per_step': n, 'algorithmic_tflop_per_step': fl / 1e12,
                'achieved': fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0}
    cls_f32, cls_split = klass('conv_mfma'), klass('conv_split')
    cls_f32['frac'] = cls_f32['achieved'] / FP32_MFMA_PEAK_TFLOPS
    cls_split['frac'] = cls_split['achieved'] / SPLIT_PEAK_TFLOPS

    if rank == 0:
        out = {
            'metric': 'micrographs/sec (4096x4096 fp32) denoise+score, NMS parity',
            'value': world * args.steps / dt,
            'unit': 'micrographs/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': 1e3 * dt / args.steps,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f32 (fp32 MFMA kernels only: TPZ_EXACT_FP32 is set)' if os.environ.get('TPZ_EXACT_FP32') else
                     'f32 (convs: fp32 operands carried as two f16 halves on the f16 MFMA, three exact products per '
                     'multiply-add accumulated in f32 -- fp32-level error, fp32-MFMA re-run on f16-range overflow; '
                     '1-channel stems, the 1-output-channel last conv and NMS in fp32)',
            'data': 'synthetic (N(0,1) micrographs seed 1000+i; seeded random weights of the named architectures)',
            'config': {
                'workload': {'pipeline': 'denoise(unet b11/t5 nf48, -s 1024 -p 500) -> score(resnet8 u64, filled) -> NMS(r=14,t=-6)',
                             'extract': 'score(resnet8 u64, filled) -> NMS(r=14,t=-6)',
                             'denoise': 'denoise(unet b11/t5 nf48, -s 1024 -p 500)'}[args.workload],
                'image': f'{args.size}x{args.size} fp32', 'images_per_rank': args.steps, 'lanes_per_gpu': args.lanes,
                'parallelism': f'one micrograph per rank x{world}; RCCL gather of pick tables',
                'picks_per_image': n_picks / max(1, len(scs)) if scs else None,
            },
            # the dominant kernel = the conv instantiation with the most time in a step (live HIP-event timing, one
            # profiled step on the kernel's own stream); `achieved` is algorithmic (fp32-equivalent) FLOP/s
            'roofline': {
                'bound': 'mfma', 'kernel': dom_name,
                'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s', 'frac': achieved / peak, 'traffic': traffic,
                'traffic_note': 'HBM bytes per launch (FETCH_SIZE + WRITE_SIZE, raw) from profiles/r01_split_head_pmc.txt; '
                                'algorithmic bytes 8.7e9',
                'peak_basis': ('f16 dense MFMA peak 2500 TFLOP/s / 3 MFMAs per fp32-equivalent MAC (2xf16 split)'
                               if is_split else 'fp32 MFMA peak'),
                'launches_per_step': dom_n, 'avg_launch_ms': dom_ms / max(1e-9, dom_n),
                'timed_over': ('the timed steps themselves (HIP events on each lane\'s stream)' if not args.no_kernel_timing
                               else 'one extra step after the timed region'),
                'algorithmic_tflop_per_launch': dom_flops / max(1e-9, dom_n) / 1e12,
                'share_of_step': dom_ms / (1e3 * dt / args.steps) if dt > 0 else None,
                'class_conv_mfma_fp32': cls_f32, 'class_conv_split_2xf16': cls_split,
                'conv_kernel_ms_per_step': conv_ms, 'conv_algorithmic_tflop_per_step': conv_flops / 1e12,
                'top_kernels': [{'kernel': k[0], 'ms': k[1], 'launches': k[2], 'tflops': k[3] / k[1] / 1e9}
                                for k in kernels[:6]],
                **({} if not args.no_kernel_timing else other),
                'coverage': ('convolution launches of >= 20 GFLOP (the rest, elementwise and NMS kernels are not timed inside '
                             'the timed region)' if not args.no_kernel_timing else 'every launch of one extra step'),
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(models, args)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
