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
shapes are those of the named configs.  Multi-GPU: every rank processes its own K images (weak scaling),
then the pick tables are gathered to rank 0 over RCCL (the only collective; timed separately as `gather_ms`).
`--gpus N` under torchrun uses the ranks it is given; run as plain `python bench.py --gpus N` it starts its own N
rank processes (topaz_amd.parallel.launch_local_ranks), one per GPU, and rank 0 prints the line.

Prints ONE JSON line (see the driver contract) including
  roofline     -- the dominant kernel (the conv instantiation with the most time in a step): algorithmic FLOP / HIP-event
                  time of its launches in the timed steps themselves; class aggregates, HBM-bound rows and the sustained-MFMA
                  probe from extra steps after the timed region
  energy       -- the same roofline in joules: board power x time per micrograph against what the step's f16 MFMA FLOP cost
                  at the probe's pJ / FLOP (the board runs the step at its power cap)
  cpu_baseline -- the oracle (CPU restatement of the reference, torch-CPU) timed on a bounded sample of the same workload on
                  one socket of this host, rank 0 and N=1 only (`--cpu-full`: one whole 4096^2 micrograph)
  parity       -- the HIP path on the arrays of that cpu_baseline leg against the oracle's outputs: |denoised pixel| and
                  |logit| deltas (bar 1e-4), device NMS on the oracle's logits (bar: identical picks); exit status 3 on failure
  configs      -- BASELINE configs 2 / 3 / 5 and the detectors users run, one by one: ms, fraction of the 2xf16 peak, launches,
                  board power and shader clock while the config runs back to back
  exact_fp32   -- the same step with every convolution pinned to the fp32-MFMA kernels (tpz_ctx_set_exact)
  leg_seconds  -- what each leg after the timed region cost (a default run takes well under a minute)
`--extras` adds the secondary legs (minutes): pcie_inclusive (the step fed from pinned host memory), the A/B legs
row_major_raster / patch_lanes_unbatched / full_patch_tensors, cli_inclusive (the CLI itself from MRC files on tmpfs) and a CPU
baseline per config.
`--dry-run` (CPU box, no GPU): skips the hot path and fabricates pick tables so that the launcher, the barriers and the
gather can be exercised over gloo; its line carries "dry_run": true and is not a measurement.
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
HBM_PEAK_BYTES = 8.0e12             # HBM3E peak (same guide); ~6.3e12 is what a copy kernel achieves


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


def _cpu_model() -> str:
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


# CPUs this process was allowed before the rank pinned itself next to its GPU (main): the CPU-baseline legs use one whole
# socket of THOSE, not the GPU's NUMA node
_ORIG_AFFINITY = None


def one_socket_cores():
    """(physical cores of ONE socket, CPU ids of their first hardware threads) from /proc/cpuinfo, restricted to the CPUs this
    process may run on -- BASELINE.md 3 / SURVEY 8(d) time the CPU path on the physical cores of one socket.  (None, None) where
    the topology cannot be read."""
    try:
        allowed = set(_ORIG_AFFINITY) if _ORIG_AFFINITY is not None else set(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        allowed = None
    cores, cur = {}, {}
    try:
        for line in list(open('/proc/cpuinfo')) + ['\n']:
            if ':' in line:
                k, v = line.split(':', 1)
                cur[k.strip()] = v.strip()
            elif cur:
                if 'processor' in cur and 'physical id' in cur and 'core id' in cur:
                    cpu = int(cur['processor'])
                    if allowed is None or cpu in allowed:
                        cores.setdefault(int(cur['physical id']), {}).setdefault(int(cur['core id']), cpu)
                cur = {}
    except (OSError, ValueError):
        return None, None
    if not cores:
        return None, None
    sock = max(cores, key=lambda k: len(cores[k]))
    return len(cores[sock]), sorted(cores[sock].values())


class OneSocket:
    """torch's intra-op pool on the physical cores of one socket for the CPU-baseline legs (the main thread and the pool's
    workers are pinned to them; restored afterwards).  `.cores` = the threads actually used."""

    def __enter__(self):
        self.prev_threads = torch.get_num_threads()
        self.prev_aff = os.sched_getaffinity(0) if hasattr(os, 'sched_getaffinity') else None
        n, cpus = one_socket_cores()
        self.cores = n or self.prev_threads
        self.pinned = False
        if n:
            torch.set_num_threads(n)
            try:
                from topaz_amd.parallel import _set_affinity_all_threads
                _set_affinity_all_threads(cpus)
                self.pinned = True
            except OSError:
                pass
        return self

    def __exit__(self, *exc):
        torch.set_num_threads(self.prev_threads)
        if self.pinned and self.prev_aff is not None:
            try:
                from topaz_amd.parallel import _set_affinity_all_threads
                _set_affinity_all_threads(sorted(self.prev_aff))
            except OSError:
                pass


def cpu_baseline(models, args, keep=None):
    """The oracle (torch-CPU convs, C NMS) timed on this host.  `keep` (a dict) receives the leg's own inputs and outputs -- the
    patch it denoised, the crop it scored, the logits, the picks -- for the parity check of the GPU path (parity_vs_oracle).
    Default: a bounded sample of the step's own units of work --
    the denoiser on ONE full patch of the -s/-p tiling (2024^2 at the defaults; a 4096^2 micrograph is 16 such crops, 3.0x the
    image), the scorer and the suppression on a --cpu-sample crop (2048^2) -- each scaled by the pixels the
    whole micrograph pushes through that stage.  (Round 2 timed everything on a 1024^2 crop: oneDNN runs small images at a
    lower rate, which read 25 % slow against the whole micrograph.)  --cpu-full: micrograph 0 itself, nothing scaled (minutes)."""
    from oracle import denoising as oden
    from oracle import nms as onms
    from oracle import scoring as oscoring
    full = args.cpu_full
    img = np.random.RandomState(1000).randn(args.size, args.size).astype(np.float32)
    with OneSocket() as sock:
        return _cpu_baseline(models, args, full, img, sock.cores, sock.pinned, keep)


def _cpu_baseline(models, args, full, img, threads, pinned, keep=None):
    from oracle import denoising as oden
    from oracle import nms as onms
    from oracle import scoring as oscoring
    per_image = 0.0
    parts = {}
    full_px = float(args.size) ** 2
    S = args.size if full else min(args.size, args.cpu_sample)
    x = img[:S, :S].copy()
    # (one small untimed call per network: primitive creation / thread pool start-up are not the workload)
    if 'denoise' in models:
        sd = models['denoise'][1]
        oden.denoise('unet', sd, img[:256, :256].copy(), -1)
        if full:
            t0 = time.time(); x = oden.denoise('unet', sd, x, args.patch_size, args.patch_padding); t = time.time() - t0
            per_image += t
        else:
            P = min(args.size, args.patch_size + 2 * args.patch_padding)
            t0 = time.time(); x = oden.denoise('unet', sd, img[:P, :P].copy(), -1); t = time.time() - t0
            if keep is not None:
                keep['denoise_in'], keep['denoise_out'] = img[:P, :P].copy(), x.copy()
            S = P                              # the later stages see what they see in the step: a denoised image
            # pixels the full job pushes through the net with -s/-p patching
            n_px = 0
            for i in range(0, args.size, args.patch_size):
                for j in range(0, args.size, args.patch_size):
                    h = min(args.size, i + args.patch_size + args.patch_padding) - max(0, i - args.patch_padding)
                    w = min(args.size, j + args.patch_size + args.patch_padding) - max(0, j - args.patch_padding)
                    n_px += h * w
            per_image += t * n_px / float(P * P)
            parts['denoise_patch'] = f'{P}x{P}'
        parts['denoise_s'] = round(t, 3)
    if 'score' in models:
        sd = models['score'][1]
        oscoring.score('resnet8', sd, img[:256, :256].copy())
        if not full and S > args.cpu_score_sample:
            S = args.cpu_score_sample          # (bounds the leg on one socket's cores; 1536^2 runs at the large-image rate)
            x = x[:S, :S].copy()
        t0 = time.time(); logit = oscoring.score('resnet8', sd, x); t = time.time() - t0
        parts['score_s'] = round(t, 3)
        per_image += t if full else t * full_px / (S * S)
        t0 = time.time(); o_sc, o_co = onms.nms2d(logit, args.radius, args.threshold); t = time.time() - t0
        if keep is not None:
            keep['score_in'], keep['logit'], keep['picks'] = x.copy(), logit.copy(), (o_sc, o_co)
        parts['nms_s'] = round(t, 3)
        per_image += t if full else t * full_px / (S * S)
    sample = (f'micrograph 0 at {S}x{S} (the whole workload of one step, nothing scaled), times {parts}' if full else
              f'one patch of the denoise tiling of micrograph 0 ({parts.get("denoise_patch", "-")}) through the oracle U-Net, a '
              f'{S}x{S} crop of its result through the oracle scorer + C NMS, times {parts}, each stage scaled by the pixels a '
              f'{args.size}x{args.size} micrograph pushes through it')
    return {'value': 1.0 / per_image, 'unit': 'micrographs/s', 'cores': threads, 'kind': 'port',
            'cores_are': 'the physical cores of one socket (one torch intra-op thread per core, pinned)' if pinned else
                         'torch intra-op threads (socket topology not readable: unpinned)',
            'cpu': _cpu_model(), 'host_logical_cpus': os.cpu_count(),
            'sample': sample + '; oracle = torch-CPU fp32 convs (oneDNN), C NMS'}


def parity_vs_oracle(models, keep, args, dev):
    """The GPU path on the very arrays the cpu_baseline leg pushed through the oracle: the patch it
    denoised, the crop it scored (the ORACLE's denoised pixels, so each stage is compared on identical input), its logits
    through the device NMS.  Bars: 1e-4 absolute on denoised pixels and logits (north_star), identical pick tables."""
    from topaz_amd import runtime as rt
    out = {'atol': 1e-4}
    if 'denoise_in' in keep and 'denoise' in models:
        y = models['denoise'][0].denoise_device(torch.from_numpy(keep['denoise_in']).to(dev), -1, 0).cpu().numpy()
        out['denoise_max_abs'] = float(np.abs(y - keep['denoise_out']).max())
        out['denoise_on'] = 'x'.join(map(str, keep['denoise_in'].shape)) + ' patch of micrograph 0 (one _denoise call)'
    if 'logit' in keep and 'score' in models:
        lg = models['score'][0](torch.from_numpy(keep['score_in']).to(dev)[None, None])[0, 0]
        out['logit_max_abs'] = float(np.abs(lg.cpu().numpy() - keep['logit']).max())
        out['logit_range'] = [float(keep['logit'].min()), float(keep['logit'].max())]
        out['logit_on'] = 'x'.join(map(str, keep['score_in'].shape)) + " crop of the oracle's denoised patch"
        sc, co = rt.nms(torch.from_numpy(keep['logit']).to(dev), args.radius, args.threshold)
        o_sc, o_co = keep['picks']
        out['picks'] = int(len(o_sc))
        out['picks_equal'] = bool(len(o_sc) == sc.numel() and np.array_equal(co.cpu().numpy(), o_co)
                                  and np.array_equal(sc.cpu().numpy(), o_sc))
        # and the chain as a user sees it: picks of the GPU's own logits against the oracle's (a pick may differ where two
        # competing scores lie within 2e-4 of each other)
        sc2, co2 = rt.nms(lg, args.radius, args.threshold)
        a = {tuple(c) for c in co2.cpu().numpy().tolist()}
        b = {tuple(c) for c in np.asarray(o_co).tolist()}
        out['picks_differing_end_to_end'] = int(len(a ^ b))
    out['ok'] = bool(out.get('denoise_max_abs', 0.0) <= 1e-4 and out.get('logit_max_abs', 0.0) <= 1e-4 and
                     out.get('picks_equal', True))
    out['what'] = ('HIP path vs oracle on the arrays of the cpu_baseline leg: |denoised pixel| and |logit| deltas (bar 1e-4), the '
                   "device NMS on the oracle's own logit map (bar: identical coordinates and scores)")
    return out


def kernel_source_sha1() -> str:
    import hashlib
    try:
        return hashlib.sha1(open(os.path.join(ROOT, 'topaz_amd', 'csrc', 'conv_split.h'), 'rb').read()).hexdigest()
    except OSError:
        return ''


def measured_traffic(dom_name: str, args):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/pmc_dominant.json, written by
    tools/pmc_traffic.py from separate `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE` runs of this same bench command,
    corrected as MI355X_MICROARCH.md prescribes).  PMC counters cannot be read from inside the process, so the
    figure is only reported when that file describes exactly this kernel and image size -- otherwise null."""
    path = os.path.join(ROOT, 'profiles', 'pmc_dominant.json')
    try:
        rec = json.load(open(path))
    except (OSError, ValueError):
        return None, None
    if rec.get('kernel') != dom_name or rec.get('size') != args.size or rec.get('workload') != args.workload:
        return None, None
    # ... and was profiled on THIS kernel source: the file records the sha1 of csrc/conv_split.h at the profiled commit
    if rec.get('conv_split_h_sha1') != kernel_source_sha1():
        return None, {'stale': True, 'profiled_at_commit': rec.get('profiled_at_commit'),
                      'reason': 'profiles/pmc_dominant.json was collected on another version of csrc/conv_split.h '
                                '(re-run tools/pmc_traffic.py)'}
    return rec.get('traffic_bytes_per_launch'), {k: rec.get(k) for k in (
        'fetch_bytes_raw', 'fetch_bytes_corrected', 'write_bytes', 'algorithmic_bytes', 'source', 'correction',
        'profiled_at_commit', 'conv_split_h_sha1')}


def dry_run(args, rank, world):
    """CPU-only plumbing check (no hot path, not a measurement): fabricated pick tables through the same barriers,
    max-over-ranks reduction and gather as the real run, over gloo."""
    from topaz_amd import parallel
    dev = torch.device('cpu')
    ids = [rank + i * world for i in range(args.steps)]
    scs, cds = [], []
    for i in ids:
        rs = np.random.RandomState(100 + i)
        n = int(rs.randint(1, 50))
        scs.append(torch.from_numpy(np.sort(rs.randn(n).astype(np.float32))[::-1].copy()))
        cds.append(torch.from_numpy(rs.randint(0, args.size, size=(n, 2)).astype(np.int32)))
    parallel.barrier(dev)
    t0 = time.perf_counter()
    tables = parallel.gather_pick_tables(ids, scs, cds, dev) if world > 1 else {i: None for i in ids}
    parallel.barrier(dev)
    dt = parallel.max_over_ranks(time.perf_counter() - t0, dev)
    total = int(parallel.sum_over_ranks(float(args.steps), dev))
    rccl_world = int(parallel.sum_over_ranks(1.0, dev))
    rank_ms = parallel.gather_scalars(1e3 * dt / max(1, args.steps), dev)
    if rank == 0:
        assert sorted(tables) == list(range(total)), sorted(tables)
        print(json.dumps({'dry_run': True, 'metric': 'none (plumbing check, no hot path)', 'value': None, 'n_gpus': world,
                          'steps': args.steps, 'warmup': args.warmup, 'images_gathered': len(tables), 'scaling': args.scaling,
                          'rccl_world': rccl_world,
            'host_placement': {'rank0_cpus': len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else None,
                               'pinned_to_gpu_numa_node': bool(os.environ.get('TOPAZ_AMD_RANK_CPUS')),
                               'host_cpus_before_pinning': len(_ORIG_AFFINITY) if _ORIG_AFFINITY is not None else None}, 'rank_ms_per_step': {'min': min(rank_ms), 'max': max(rank_ms)},
                          'ms_per_step': 1e3 * dt / max(1, args.steps), 'backend': 'gloo'}))
    if world > 1:
        torch.distributed.destroy_process_group()


def pcie_inclusive(models, host_imgs, args, dev, n=4):
    """The same step with the micrograph starting in host memory and the pick table ending there, through the staging
    ring of the C-ABI (tpz_stage: pinned slot + device slot + copy stream): the H2D copy of micrograph i+1 is queued before
    the kernels of micrograph i, the picks come back with a synchronous D2H (NMS synchronises anyway to learn the pick
    count).  File I/O excluded."""
    from topaz_amd import runtime as rt
    S = args.size
    ctx = rt.get_context(dev.index)
    stage = rt.Stage(ctx, S * S * 4, depth=2)

    def submit(i):
        np.copyto(stage.host_array(i % 2, (S, S)), host_imgs[i % len(host_imgs)])     # "read the file into pinned memory"
        stage.upload(i % 2, S * S * 4)

    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    submit(0)
    for i in range(n):
        stage.acquire(i % 2)
        if i + 1 < n:
            submit(i + 1)                       # slot (i + 1) % 2 was released by step i - 1
        sc, co = run_step(models, stage.device_tensor(i % 2, (S, S)), args)
        stage.release(i % 2)
        sc.cpu()
        if co is not None:
            co.cpu()
    torch.cuda.synchronize(dev)
    t = time.perf_counter() - t0
    stage.close()
    return {'value': n / t, 'ms_per_step': 1e3 * t / n, 'steps': n, 'unit': 'micrographs/s',
            'note': 'input from host memory through tpz_stage (H2D of the next micrograph under the compute of this one), '
                    'pick table copied back to the host; file I/O excluded'}


class GpuSampler:
    """Shader clock and socket power of THIS rank's GPU while the timed region runs, from the amdgpu hwmon files of the DRM card
    with the HIP device's PCI address (freq1_input: sclk in Hz; power1_input / power1_average: microwatts), sampled by a thread
    every 25 ms: the boxes of the pool sustain different clocks under the f16-MFMA load (7 % spread of the step time,
    profiles/r03_bench_boxes.txt), so a slow box can be told from a regression.  Fields are null where the files do not exist."""

    def __init__(self, index: int = 0):
        import glob
        self.sclk, self.power = [], []
        self._stop = False
        self._thread = None
        self.dev, self.fq, self.pw = None, None, None
        try:
            pr = torch.cuda.get_device_properties(index)
            bdf = f'{int(getattr(pr, "pci_domain_id", 0)):04x}:{int(pr.pci_bus_id):02x}:{int(pr.pci_device_id):02x}.0'
        except (AttributeError, RuntimeError, AssertionError):
            bdf = None
        for d in sorted(glob.glob('/sys/class/drm/card[0-9]*/device')):
            if bdf and os.path.basename(os.path.realpath(d)) == bdf:
                self.dev = d
        if self.dev:
            for name in ('freq1_input',):
                hits = glob.glob(os.path.join(self.dev, 'hwmon', 'hwmon*', name))
                self.fq = hits[0] if hits else None
            for name in ('power1_input', 'power1_average'):
                hits = glob.glob(os.path.join(self.dev, 'hwmon', 'hwmon*', name))
                if hits:
                    self.pw = hits[0]
                    break

    def _read(self):
        for path, out, unit in ((self.fq, self.sclk, 1e6), (self.pw, self.power, 1e6)):
            if path:
                try:
                    out.append(float(open(path).read()) / unit)
                except (OSError, ValueError):
                    pass

    def _run(self):
        while not self._stop:
            self._read()
            time.sleep(0.025)

    def __enter__(self):
        if self.fq or self.pw:
            import threading
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop = True
        if self._thread:
            self._thread.join()

    def summary(self):
        mean = lambda v: round(sum(v) / len(v), 1) if v else None
        return {'sclk_mhz_mean': mean(self.sclk), 'sclk_mhz_min': min(self.sclk) if self.sclk else None,
                'power_w_mean': mean(self.power), 'power_w_max': max(self.power) if self.power else None,
                'samples': max(len(self.sclk), len(self.power)),
                'source': (f'{self.dev}/hwmon (freq1_input, power1_input)' if self.dev else 'no DRM card with the HIP device\'s PCI address') +
                          ', 25 ms sampling during the timed region'}


# the reference's own layer FLOP (2 * Cout * Cin * k^dims * output pixels, every patch / tile in full) of the BASELINE configs
CONFIG_TFLOP = {'c2_extract_resnet8_u64': 44.23, 'c2_extract_resnet8_u32': 11.09, 'c3_denoise_unet_patched': 29.06,
                'c3_denoise_unet_whole': 9.68, 'c5_denoise3d_unet3d': 516.7,
                # not BASELINE configs, but what users run: `topaz extract` defaults to -m resnet16 (commands/extract.py:16-53,
                # factory.py:36-51 = units 64; resnet16_u32 is the shipped blob) and north_star names Conv127 (factory.py:15-17);
                # TFLOP per 4096^2 micrograph from SURVEY.md 8(d)
                'extract_resnet16_u64': 61.99, 'extract_resnet16_u32': 15.53, 'extract_conv127_u32': 3.61}


def baseline_configs(ctx, models, imgs, args, dev, with_cpu):
    """BASELINE.json configs 2, 3 and 5 one by one, after the timed region: ms per unit (best of two after
    one warm-up), the reference's algorithmic TFLOP, the executed TFLOP (what the launches really compute: patch windows,
    per-parity decoders), executed TFLOP/s against the 2xf16 ceiling of 833, launches, and the oracle on a bounded sample of
    the same config on this host."""
    from oracle import denoising as oden
    from oracle import nms as onms
    from oracle import scoring as oscoring
    from tools import synth_weights as sw
    from topaz_amd import runtime as rt
    from topaz_amd.denoise import Denoise3D
    from topaz_amd.denoising.models import DenoiseNet
    from topaz_amd.model.factory import load_model
    x = imgs[0]
    S = args.size
    out = {}

    def measure(key, fn, n=2):
        fn()
        torch.cuda.synchronize(dev)
        best = None
        for _ in range(n):
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize(dev)
            t = time.perf_counter() - t0
            best = t if best is None else min(best, t)
        ctx.prof_enable(1); ctx.prof_reset()
        l0 = ctx.launches()
        fn()
        torch.cuda.synchronize(dev)
        launches = ctx.launches() - l0
        _, _, fl = ctx.prof_get(0)
        ctx.prof_enable(False)
        # board power and shader clock while the config runs back to back for ~0.5 s: which configs sit at the power cap
        # (their time is their energy) and which leave power on the table (their time is stalls)
        with GpuSampler(dev.index or 0) as gs:
            t_end = time.perf_counter() + 0.5
            while time.perf_counter() < t_end:
                fn()
            torch.cuda.synchronize(dev)
        g = gs.summary()
        out[key] = {'ms': 1e3 * best, 'algorithmic_tflop': CONFIG_TFLOP[key], 'executed_tflop': fl / 1e12,
                    'executed_tflops': fl / best / 1e12, 'frac': fl / best / 1e12 / SPLIT_PEAK_TFLOPS,
                    'reference_tflops': CONFIG_TFLOP[key] / best, 'conv_and_elementwise_launches': launches,
                    'power_w_mean': g['power_w_mean'], 'sclk_mhz_mean': g['sclk_mhz_mean']}
        return out[key]

    def scorer(m):
        return lambda: rt.nms(m(x[None, None])[0, 0], args.radius, args.threshold)

    u32 = load_model('resnet8_u32')
    u32.eval(); u32.fill(); u32.cuda()
    m64 = models['score'][0] if 'score' in models else sw.hip_resnet('resnet8', 64, seed=7)[0]
    sd64 = models['score'][1] if 'score' in models else None
    dn = models['denoise'][0] if 'denoise' in models else None
    measure('c2_extract_resnet8_u64', scorer(m64))
    measure('c2_extract_resnet8_u32', scorer(u32))
    # the detectors users actually run: the CLI default resnet16 (u64, seeded: blob missing upstream), the
    # shipped resnet16_u32, conv127 (u32, BN + PReLU, seeded); each with its three heaviest kernels
    from topaz_amd.model.classifier import LinearClassifier
    m16, sd16 = sw.hip_resnet('resnet16', 64, seed=7)
    r16u32 = load_model('resnet16_u32')
    r16u32.eval(); r16u32.fill(); r16u32.cuda()
    sd127 = sw.basic_sd((7, 5, 5, 5, 5), 32, 7)
    c127 = LinearClassifier('conv127', sd127)
    c127.eval(); c127.fill(); c127.cuda()
    for key, m in (('extract_resnet16_u64', m16), ('extract_resnet16_u32', r16u32), ('extract_conv127_u32', c127)):
        measure(key, scorer(m))
        rows = sorted(ctx.prof_kernels(), key=lambda r: -r[1])[:3]        # (the profiled call of measure(): every launch timed)
        out[key]['top_kernels'] = [{'kernel': nm, 'ms': ms, 'launches': n, 'tflops': fl / ms / 1e9,
                                    'frac': fl / ms / 1e9 / SPLIT_PEAK_TFLOPS} for nm, ms, n, fl in rows if ms > 0]
        n_conv, n_split, off = m.device_model.split_layers()
        out[key]['layers_on_2xf16_path'] = f'{n_split} of {n_conv}'
    if dn is not None:
        measure('c3_denoise_unet_patched', lambda: dn.denoise_device(x, args.patch_size, args.patch_padding))
        measure('c3_denoise_unet_whole', lambda: dn.denoise_device(x, -1, 0))
    if dn is not None:
        # (not a BASELINE config: the pretrained fully convolutional denoiser, 11x11 64->64 body -- VERDICT r03 item 6)
        from topaz_amd.denoise import Denoise
        CONFIG_TFLOP['denoise_fcnn_patched'] = 17.15 * 50.4 / 16.8
        fc = Denoise('fcnn')
        measure('denoise_fcnn_patched', lambda: fc.denoise_device(x, args.patch_size, args.patch_padding))
        out['denoise_fcnn_patched']['unit'] = f'one {S}x{S} micrograph, -s {args.patch_size} -p {args.patch_padding}'
    sd3 = sw.unet_sd(13, nf=48, base_width=7, top_width=3, dims=3)
    d3 = Denoise3D(DenoiseNet('unet-3d', sd3))
    tomo = torch.randn(256, 512, 512, device=dev, generator=torch.Generator(device=dev).manual_seed(2000))
    measure('c5_denoise3d_unet3d', lambda: d3.model.device_model.denoise_3d(tomo, 96, 48), n=1)
    out['c5_denoise3d_unet3d']['unit'] = 'one 512x512x256 tomogram, 96/48 tiles (108 tiles of 192^3)'
    del tomo
    for k in ('c2_extract_resnet8_u64', 'c2_extract_resnet8_u32', 'c3_denoise_unet_patched', 'c3_denoise_unet_whole',
              'extract_resnet16_u64', 'extract_resnet16_u32', 'extract_conv127_u32'):
        if k in out:
            out[k]['unit'] = f'one {S}x{S} micrograph'
    def cpu_legs(sock):
        # the oracle on a bounded sample of each config, scaled by the pixels / voxels the whole unit pushes through the net,
        # on the physical cores of one socket (OneSocket)
        P = 1024
        crop = np.random.RandomState(1000).randn(P, P).astype(np.float32)
        full_px = float(S) * S

        def cpu(key, seconds, scale, sample):
            if key in out:
                out[key]['cpu_baseline'] = {'ms': 1e3 * seconds * scale, 'kind': 'port', 'cores': sock.cores,
                                            'sample': sample, 'gpu_over_cpu': seconds * scale / (out[key]['ms'] * 1e-3)}
        sdu = {k: v.numpy() for k, v in u32.state_dict().items()}
        oscoring.score('resnet8', sdu, crop[:256, :256].copy())
        t0 = time.time(); lg = oscoring.score('resnet8', sdu, crop); onms.nms2d(lg, args.radius, args.threshold); t = time.time() - t0
        cpu('c2_extract_resnet8_u32', t, full_px / (P * P), f'{P}x{P} crop through the oracle scorer + C NMS, scaled by pixels')
        if sd64 is not None:
            t0 = time.time(); lg = oscoring.score('resnet8', sd64, crop); onms.nms2d(lg, args.radius, args.threshold); t = time.time() - t0
            cpu('c2_extract_resnet8_u64', t, full_px / (P * P), f'{P}x{P} crop through the oracle scorer + C NMS, scaled by pixels')
        for key, arch, sdx in (('extract_resnet16_u64', 'resnet16', sd16),
                               ('extract_resnet16_u32', 'resnet16', {k: v.numpy() for k, v in r16u32.state_dict().items()}),
                               ('extract_conv127_u32', 'conv127', sd127)):
            oscoring.score(arch, sdx, crop[:256, :256].copy())
            t0 = time.time(); lg = oscoring.score(arch, sdx, crop); onms.nms2d(lg, args.radius, args.threshold); t = time.time() - t0
            cpu(key, t, full_px / (P * P), f'{P}x{P} crop through the oracle scorer + C NMS, scaled by pixels')
        if dn is not None:
            sd = models['denoise'][1]
            oden.denoise('unet', sd, crop[:256, :256].copy(), -1)
            t0 = time.time(); oden.denoise('unet', sd, crop, -1); t = time.time() - t0
            n_px = 0
            for i in range(0, S, args.patch_size):
                for j in range(0, S, args.patch_size):
                    n_px += ((min(S, i + args.patch_size + args.patch_padding) - max(0, i - args.patch_padding)) *
                             (min(S, j + args.patch_size + args.patch_padding) - max(0, j - args.patch_padding)))
            cpu('c3_denoise_unet_patched', t, n_px / float(P * P), f'{P}x{P} crop through the oracle U-Net, scaled to the '
                f'{n_px} pixels of the 16 padded patches the reference computes')
            cpu('c3_denoise_unet_whole', t, full_px / (P * P), f'{P}x{P} crop through the oracle U-Net, scaled by pixels')
        T = 96
        vol = torch.from_numpy(np.random.RandomState(2000).randn(T, T, T).astype(np.float32))
        tsd3 = oden.to_torch_sd(sd3)
        oden.denoise_whole('unet-3d', tsd3, vol[:32, :32, :32].clone())
        t0 = time.time(); oden.denoise_whole('unet-3d', tsd3, vol); t = time.time() - t0
        cpu('c5_denoise3d_unet3d', t, 108.0 * 192 ** 3 / float(T ** 3), f'one {T}^3 tile through the oracle 3-D U-Net, scaled to the '
            '108 tiles of 192^3 voxels the reference computes')

    if with_cpu:
        with OneSocket() as sock:
            cpu_legs(sock)
    return out


def blob_micrograph(S: int, seed: int, spacing: int = 36, sigma: float = 6.0, amp: float = 2.5, jitter: int = 5):
    """N(0,1) noise over a jittered grid of dark Gaussian blobs ("particles"): after denoising, the pretrained detector finds
    ~1 100 picks per Mpx at r = 14, t = -6 (oracle chain on a 640^2 crop: 466 picks for 256 blobs), so that the extract half
    of the CLI leg really produces, gathers and writes a pick table (pure noise denoises to a flat map: 0 picks)."""
    rs = np.random.RandomState(seed)
    x = rs.randn(S, S).astype(np.float32)
    R = int(4 * sigma)
    yy, xx = np.mgrid[-R:R + 1, -R:R + 1].astype(np.float32)
    stamp = (amp * np.exp(-(yy ** 2 + xx ** 2) / (2 * sigma ** 2))).astype(np.float32)
    n = 0
    for cy in range(spacing // 2 + R, S - R - spacing // 2, spacing):
        for cx in range(spacing // 2 + R, S - R - spacing // 2, spacing):
            y, xq = cy + rs.randint(-jitter, jitter + 1), cx + rs.randint(-jitter, jitter + 1)
            x[y - R:y + R + 1, xq - R:xq + R + 1] -= stamp
            n += 1
    return x, n


def cli_inclusive(args, dev, n=16):
    """File-I/O-inclusive throughput of the CLI itself: n synthetic 4096^2 fp32 MRC files on tmpfs ->
    `topaz denoise` (pretrained unet-v0.2.1, -s 1024 -p 500, MRC out) -> `topaz extract` (pretrained resnet8_u32, r = 14) on the
    denoised files, run in this process through topaz_amd.main (model loading included, interpreter start-up not; the second
    invocation is the one reported, the first also pays the growth of the workspace pools).  Reading /
    decoding micrograph i+1 and writing micrograph i-1 overlap the GPU work of micrograph i (extract.ImageFeed, denoise._run_jobs).
    `gpu_only_ms` is the same two networks on a resident micrograph: what the files cost is the difference.  (These are the
    pretrained detectors the CLI ships; `value` above runs the heavier default architectures on seeded weights.)"""
    import contextlib
    import io
    import shutil
    import tempfile
    from topaz_amd import main as tmain
    from topaz_amd import runtime as rt
    from topaz_amd.denoise import Denoise
    from topaz_amd.model.factory import load_model
    from topaz_amd.utils.image import save_image
    S = args.size
    base = '/dev/shm' if os.path.isdir('/dev/shm') and os.access('/dev/shm', os.W_OK) else None
    d = tempfile.mkdtemp(prefix='tpz_bench_', dir=base)
    try:
        src = []
        for i in range(n):
            p = os.path.join(d, f'mic_{i:03d}.mrc')
            save_image(blob_micrograph(S, 3000 + i)[0], p)
            src.append(p)
        den_dir = os.path.join(d, 'den')
        sink = io.StringIO()
        den = [os.path.join(den_dir, os.path.basename(p)) for p in src]

        def both():
            torch.cuda.synchronize(dev)
            a = time.perf_counter()
            tmain.main(['denoise', '-m', 'unet-v0.2.1', '-s', str(args.patch_size), '-p', str(args.patch_padding), '-o', den_dir] + src)
            torch.cuda.synchronize(dev)
            b = time.perf_counter()
            tmain.main(['extract', '-m', 'resnet8_u32', '-r', str(args.radius), '-t', str(args.threshold), '-o', os.path.join(d, 'picks.txt')] + den)
            torch.cuda.synchronize(dev)
            return a, b, time.perf_counter()
        with contextlib.redirect_stderr(sink), contextlib.redirect_stdout(sink):
            c0, c1, c2 = both()          # first invocation in the process: also grows the workspace pools and the staging rings
            t0, t1, t2 = both()
        n_picks = sum(1 for _ in open(os.path.join(d, 'picks.txt'))) - 1
        if S >= 2048 and n_picks <= 10000 * n * (S / 4096.0) ** 2:
            raise RuntimeError(f'cli_inclusive: only {n_picks} picks over {n} micrographs -- the extract half of the leg is not '
                               'exercising NMS output, the gather and the writer')
        # the same networks on a resident micrograph
        x = torch.from_numpy(blob_micrograph(S, 3000)[0]).to(dev)
        tl = time.perf_counter()
        dn = Denoise('unet-v0.2.1')
        m = load_model('resnet8_u32')
        m.eval(); m.fill(); m.cuda()
        torch.cuda.synchronize(dev)
        load_ms = 1e3 * (time.perf_counter() - tl)          # (what each of the two invocations pays once, whatever the file count)

        def step():
            y = dn.denoise_device(x, args.patch_size, args.patch_padding)
            return rt.nms(m(y[None, None])[0, 0], args.radius, args.threshold)
        step()
        torch.cuda.synchronize(dev)
        t3 = time.perf_counter()
        for _ in range(3):
            step()
        torch.cuda.synchronize(dev)
        gpu_ms = 1e3 * (time.perf_counter() - t3) / 3
        mb = S * S * 4 / 1e6
        return {'value': n / (t2 - t0), 'unit': 'micrographs/s', 'files': n, 'ms_per_micrograph': 1e3 * (t2 - t0) / n,
                'denoise_ms_per_micrograph': 1e3 * (t1 - t0) / n, 'extract_ms_per_micrograph': 1e3 * (t2 - t1) / n,
                'first_invocation_ms_per_micrograph': 1e3 * (c2 - c0) / n,
                'gpu_only_ms_per_micrograph': gpu_ms, 'model_loading_ms_per_job': load_ms,
                # the per-job fixed cost (loading and packing both models: once per invocation whatever the file count) taken out
                'ms_per_micrograph_without_model_loading': (1e3 * (t2 - t0) - load_ms) / n,
                'vs_gpu_only': (1e3 * (t2 - t0) - load_ms) / n / gpu_ms if gpu_ms > 0 else None,
                'picks': n_picks,
                'picks_per_micrograph': n_picks / n,
                'input': 'N(0,1) noise over a jittered 36-px grid of dark Gaussian blobs (sigma 6, amplitude 2.5): particles for the '
                         'pretrained detector to find after denoising',
                'io_mb_per_micrograph': {'read': 2 * mb, 'written': mb},
                # file traffic this ONE rank sustains through both commands (eight ranks on a host ask for eight times this)
                'io_mb_per_s': {'read': 2 * mb * n / (t2 - t0), 'written': mb * n / (t2 - t0),
                                'denoise_job': 2 * mb * n / (t1 - t0), 'extract_job': mb * n / (t2 - t1)},
                'where': d if base is None else 'tmpfs (/dev/shm)',
                'note': 'topaz denoise -m unet-v0.2.1 -> topaz extract -m resnet8_u32 through topaz_amd.main in this process, model '
                        'loading included; MRC read + decode and MRC write overlapped with the GPU work by reader / writer threads'}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def timed_steps(models, imgs, args, dev, n):
    """n steps bracketed by device synchronisation; returns (seconds, pick tables)"""
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    out = [run_step(models, imgs[i % len(imgs)], args) for i in range(n)]
    torch.cuda.synchronize(dev)
    return time.perf_counter() - t0, out


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
    ap.add_argument('--cpu-sample', type=int, default=2048)
    ap.add_argument('--cpu-score-sample', type=int, default=1536, help='cpu_baseline: crop the scorer + NMS are timed on')
    ap.add_argument('--cpu-full', action='store_true', help='cpu_baseline on one whole micrograph instead of a crop (minutes)')
    ap.add_argument('--no-kernel-timing', action='store_true', help='do not record HIP events around the launches of the timed steps')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--extras', action='store_true',
                    help='also run the secondary legs: pcie_inclusive, the A/B legs (row_major_raster, patch_lanes_unbatched, '
                         'full_patch_tensors), cli_inclusive and a CPU baseline per config (minutes)')
    ap.add_argument('--no-extras', action='store_true', help='also skip exact_fp32 (the only secondary leg of a default run)')
    ap.add_argument('--no-configs', action='store_true', help='skip the per-config legs (BASELINE configs 2, 3, 5 one by one)')
    ap.add_argument('--exact-steps', type=int, default=2)
    ap.add_argument('--dry-run', action='store_true', help='CPU-only plumbing check over gloo (no hot path, not a measurement)')
    ap.add_argument('--scaling', default='weak', choices=['weak', 'strong'],
                    help="weak (default): every rank times --steps micrographs; strong: BASELINE config 4's FIXED job of --images "
                         'micrographs is dealt i = rank (mod N) to the ranks (--steps is then ignored)')
    ap.add_argument('--images', type=int, default=256, help='--scaling strong: micrographs of the whole job')
    args = ap.parse_args()

    from topaz_amd import parallel
    if args.gpus > 1 and not parallel.under_launcher():
        # plain `python bench.py --gpus N`: be the launcher -- one rank process per GPU, rank 0 prints the line
        sys.exit(parallel.launch_local_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))
    global _ORIG_AFFINITY
    if hasattr(os, 'sched_getaffinity'):
        _ORIG_AFFINITY = sorted(os.sched_getaffinity(0))
    rank, local_rank, world = parallel.init_from_env('gloo' if args.dry_run else None)
    if world == 1 and not args.dry_run:
        # a single rank places itself next to its GPU exactly as the ranks of a multi-GPU job do (host_placement in the line)
        parallel.pin_this_rank(local_rank, 1)
    assert world == args.gpus, f'WORLD_SIZE={world} but --gpus {args.gpus}'
    if args.scaling == 'strong':
        args.steps = len(parallel.shard_indices(args.images, rank, world))       # this rank's share of the fixed job
    if args.dry_run:
        return dry_run(args, rank, world)
    # the rank's GPU (its local rank; TOPAZ_AMD_SHARE_GPU=1 + TOPAZ_AMD_DIST_BACKEND=gloo: a rehearsal of the multi-rank bench on a
    # box with fewer GPUs than ranks -- several ranks on one device, the exchange steps over gloo on host tensors) and where the
    # tensors of the collectives live (the GPU under RCCL)
    gpu = parallel.rank_device(local_rank)
    torch.cuda.set_device(gpu)
    dev = torch.device('cuda', gpu)
    cdev = parallel.collective_device(local_rank)
    from topaz_amd import runtime as rt
    ctx = rt.get_context(gpu)

    # the collective really spans the job's ranks: an all_reduce of ones over RCCL (1 without a process group)
    rccl_world = int(parallel.sum_over_ranks(1.0, cdev))
    # this rank's micrographs, resident in HBM before the timed region (global index = rank + i*world); a long strong-scaling
    # share cycles through a bounded set of resident micrographs
    n_res = min(args.steps, 8) if args.scaling == 'strong' else args.steps
    n_img = n_res + args.warmup
    host_imgs = [np.random.RandomState(1000 + rank + i * world).randn(args.size, args.size).astype(np.float32)
                 for i in range(n_img)]
    imgs = [torch.from_numpy(h).to(dev) for h in host_imgs]
    models = build_models(args.workload)
    warm_picks = [run_step(models, imgs[n_res + w], args) for w in range(args.warmup)]
    torch.cuda.synchronize(dev)
    if world > 1 and warm_picks and warm_picks[0][1] is not None:
        # the exchange step is warmed like the kernels: its collectives (the size all_gather, the gather of the packed tables)
        # connect their peers on first use -- inside the timed region that set-up would be charged to the job
        parallel.gather_pick_tables([rank + i * world for i in range(len(warm_picks))], [p[0] for p in warm_picks],
                                    [p[1] for p in warm_picks], cdev)
    del warm_picks
    if not args.no_kernel_timing:
        # roofline evidence: HIP events around the convolution launches of the TIMED steps (those of >= 20 GFLOP: ~170 of
        # ~700 launches per step, > 95 % of the kernel time), recorded on the stream the kernels are launched on and
        # resolved only after the timed region; costs < 0.5 % of the step
        ctx.prof_enable(2)
        ctx.prof_reset()

    # ---- the timed region: barrier + synchronize on both sides, exactly K steps, then the one exchange step
    torch.cuda.synchronize(dev)
    parallel.barrier(cdev)
    launches0 = ctx.launches()
    sampler = GpuSampler(gpu)
    sampler.__enter__()                                  # (joined after the timed region: its 25 ms sleep is not part of the job)
    t0 = time.perf_counter()
    cpu0 = time.process_time()                           # host CPU seconds of this rank process (all its threads)
    step_t, mem_used = [], {}
    if args.scaling == 'strong':
        # the fixed job as a SOAK through the real job path: every step's completion time (a step ends with
        # the suppression's host read of the pick count, so the host clock is the step's clock) and the device memory in use
        # after the first steps / at the end -- the workspace pools must not grow once they are warm
        picks = []
        marks = {min(args.steps, 8) - 1, args.steps // 4, args.steps // 2, args.steps - 1}
        for i in range(args.steps):
            sc_i, co_i = run_step(models, imgs[i % n_res], args)
            # the pick table leaves the device at once, as in `topaz extract` (PickSink.add takes host arrays): what is kept for
            # the final gather is its n rows, not the capacity-sized device buffers the suppression wrote them into
            picks.append((sc_i.cpu(), co_i.cpu() if co_i is not None else None))
            step_t.append(time.perf_counter())
            if i in marks:
                free_b, total_b = torch.cuda.mem_get_info(dev)
                mem_used[i] = total_b - free_b
    else:
        picks = [run_step(models, imgs[i % n_res], args) for i in range(args.steps)]
    torch.cuda.synchronize(dev)
    t_compute = time.perf_counter() - t0
    cpu_compute = time.process_time() - cpu0
    sampler._stop = True
    launches_per_step = (ctx.launches() - launches0) / max(1, args.steps)
    ids = [rank + i * world for i in range(args.steps)]
    scs, cds = [p[0] for p in picks], [p[1] for p in picks]
    have_picks = cds[0] is not None
    if have_picks and world > 1:
        parallel.gather_pick_tables(ids, scs, cds, cdev)       # RCCL: all_gather of counts + two gathers to rank 0
    torch.cuda.synchronize(dev)
    t_gather = time.perf_counter() - t0 - t_compute
    parallel.barrier(cdev)
    dt = time.perf_counter() - t0
    # every rank's own compute time per micrograph: stragglers (host contention between the ranks' launch threads, a slow
    # device) show as a spread between min and max
    sampler.__exit__()
    rank_ms = parallel.gather_scalars(1e3 * t_compute / max(1, args.steps), cdev)
    rank_cpu_ms = parallel.gather_scalars(1e3 * cpu_compute / max(1, args.steps), cdev)
    total_steps = int(parallel.sum_over_ranks(float(args.steps), cdev))
    dt = parallel.max_over_ranks(dt, cdev)
    t_gather = parallel.max_over_ranks(t_gather, cdev)
    n_picks = int(sum(int(s.numel()) for s in scs)) if have_picks else 0

    # ---- roofline: per-kernel HIP-event times of the launches of the timed steps, per step
    merged, other, conv_ms, conv_n, conv_flops = {}, {'conv_direct_ms': 0.0, 'elementwise_ms': 0.0, 'nms_ms': 0.0}, 0.0, 0, 0.0
    n_prof_steps = args.steps
    if args.no_kernel_timing:
        # no events in the timed region: time one extra step instead
        ctx.prof_enable(True)
        ctx.prof_reset()
        run_step(models, imgs[-1], args)
        torch.cuda.synchronize(dev)
        n_prof_steps = 1
    for name, ms, n, fl in ctx.prof_kernels():
        m0 = merged.setdefault(name, [0.0, 0, 0.0])
        m0[0] += ms; m0[1] += n; m0[2] += fl
    conv_ms, conv_n, conv_flops = ctx.prof_get(0)
    for k, key in ((1, 'conv_direct_ms'), (2, 'elementwise_ms'), (3, 'nms_ms')):
        other[key] += ctx.prof_get(k)[0]
    ctx.prof_enable(False)
    kernels = sorted(((nm, v[0] / n_prof_steps, v[1] / n_prof_steps, v[2] / n_prof_steps) for nm, v in merged.items()),
                     key=lambda r: -r[1])                    # (name, ms per step, launches per step, FLOP per step)
    # The denoise stage runs its patches on two concurrent streams (patch lanes): events around launches that overlap other
    # launches measure more than the kernel alone, so the per-class aggregates come from ONE extra step with the lanes off,
    # after the timed region.  The dominant kernel belongs to the scoring stage (one stream) and is taken from the timed
    # steps themselves.
    iso = {}
    if not args.no_kernel_timing and 'denoise' in models:
        ctx.set_lanes(False)
        try:
            ctx.prof_enable(2)
            ctx.prof_reset()
            run_step(models, imgs[-1], args)
            torch.cuda.synchronize(dev)
            iso = {nm: (ms, n, fl) for nm, ms, n, fl in ctx.prof_kernels()}
            conv_ms, conv_n, conv_flops = ctx.prof_get(0)
            ctx.prof_enable(False)
        finally:
            ctx.set_lanes(True)
        kernels_iso = sorted(((nm, v[0], v[1], v[2]) for nm, v in iso.items()), key=lambda r: -r[1])
    else:
        kernels_iso = kernels
    if not iso:
        conv_ms, conv_n, conv_flops = conv_ms / n_prof_steps, conv_n / n_prof_steps, conv_flops / n_prof_steps
    other = {k: v / n_prof_steps for k, v in other.items()}
    # dominant kernel: the instantiation with the most isolated time per step; its launches are timed in the timed steps
    dom_name = kernels_iso[0][0] if kernels_iso else ''
    dom_name, dom_ms, dom_n, dom_flops = next((k for k in kernels if k[0] == dom_name), ('', 0.0, 0, 0.0))
    achieved = dom_flops / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
    traffic, traffic_detail = measured_traffic(dom_name, args)
    is_split = dom_name.startswith('conv_split')
    peak = SPLIT_PEAK_TFLOPS if is_split else FP32_MFMA_PEAK_TFLOPS

    def klass(prefix):
        rows = [k for k in kernels_iso if k[0].startswith(prefix)]
        ms, n, fl = sum(k[1] for k in rows), sum(k[2] for k in rows), sum(k[3] for k in rows)
        return {'kernel_ms_per_step': ms, 'launches_per_step': n, 'algorithmic_tflop_per_step': fl / 1e12,
                'achieved': fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0}
    # what the matrix pipes of THIS box sustain: a register-resident loop of the 2xf16 kernels' MFMA on every SIMD, after the
    # timed region (profiles/r04_mfma_sustained.txt: with full-entropy operands the board holds ~1.9 GHz, not the 2.4 GHz of the
    # dense peak; with zero operands the same loop reaches the nominal rate)
    sustained = None
    probe_power = None
    if rank == 0 and is_split:
        with GpuSampler(gpu) as ps:
            tf_rand, clk_rand = ctx.mfma_sustained(300, False)
        probe_power = ps.summary().get('power_w_mean')
        tf_zero, clk_zero = ctx.mfma_sustained(300, True)
        sustained = {
            'f16_dense_tflops': tf_rand, 'fp32_equivalent_tflops': tf_rand / 3.0,
            'frac_of_sustained': achieved / (tf_rand / 3.0) if tf_rand > 0 else None,
            'zero_operand_f16_dense_tflops': tf_zero,
            'shader_clock_vs_zero_operand_run': clk_rand / clk_zero if clk_zero > 0 else None,
            'what': 'tpz_prof_mfma_sustained: v_mfma_f32_16x16x32_f16 back to back from registers, two waves per SIMD on every CU, '
                    '0.3 s, operands uniform in [-1, 1] (zero_operand: all zero); no memory traffic.  `frac` above stays against '
                    'the nominal peak; this is the ceiling the board power management leaves for that instruction',
        }
    # ---- the ENERGY view of the same roofline.  The board runs this step at its power cap from the first launch to the last
    # (gpu_clock_power.power_w_mean), so a step's time is its energy / the cap: what the matrix pipes alone need for the step's
    # executed f16 MFMA FLOP (at the pJ / FLOP of the register-resident probe, which runs at the same cap) is the floor
    energy = None
    gp = sampler.summary()
    if sustained and probe_power and gp.get('power_w_mean') and tf_rand > 0:
        j_step = gp['power_w_mean'] * dt / max(1, args.steps)
        pj_per_flop = probe_power / (tf_rand * 1e12) * 1e12
        split_flops = sum(k[3] for k in kernels if k[0].startswith('conv_split'))       # fp32-equivalent, per step, executed
        j_floor = 3.0 * split_flops * pj_per_flop * 1e-12
        energy = {'joules_per_micrograph': j_step, 'mfma_floor_joules': j_floor, 'frac': j_floor / j_step if j_step > 0 else None,
                  'probe_pj_per_f16_flop': pj_per_flop, 'probe_power_w': probe_power,
                  'f16_mfma_tflop_per_step': 3.0 * split_flops / 1e12,
                  'what': 'joules = mean board power of the timed region x time per step; floor = the step\'s executed f16 MFMA FLOP '
                          '(3 per fp32-equivalent FLOP of the conv_split launches timed in the step) x the energy per FLOP of '
                          'tpz_prof_mfma_sustained (MFMAs from registers, random operands, same power cap).  On a power-capped '
                          'board frac is the fraction of the step\'s energy -- hence of its time -- that the arithmetic itself needs'}
    cls_f32, cls_split = klass('conv_mfma'), klass('conv_split')
    cls_f32['frac'] = cls_f32['achieved'] / FP32_MFMA_PEAK_TFLOPS
    cls_split['frac'] = cls_split['achieved'] / SPLIT_PEAK_TFLOPS

    # ---- the kernels that are HBM-bound rather than MFMA-bound (arithmetic intensity below the ridge of their MFMA peak:
    # 1-channel stems, the 1-output-channel last conv and its shiftsum): algorithmic bytes / HIP-event time of ONE extra
    # step with every launch timed and the patch lanes off, against the 8 TB/s of HBM3E
    hbm_rows = []
    if rank == 0 and not args.no_kernel_timing:
        ctx.set_lanes(False)
        try:
            ctx.prof_enable(1)
            ctx.prof_reset()
            run_step(models, imgs[-1], args)
            torch.cuda.synchronize(dev)
            for nm, ms, n, fl, by in ctx.prof_kernels_bytes():
                if by <= 0 or ms <= 0:
                    continue
                pk = SPLIT_PEAK_TFLOPS if nm.startswith('conv_split') else FP32_MFMA_PEAK_TFLOPS
                if fl / by < pk * 1e12 / HBM_PEAK_BYTES:          # below the ridge: the roof is HBM
                    hbm_rows.append({'kernel': nm, 'ms': ms, 'launches': n, 'algorithmic_gb': by / 1e9,
                                     'achieved_gb_s': by / (ms * 1e-3) / 1e9, 'frac': by / (ms * 1e-3) / HBM_PEAK_BYTES,
                                     'flop_per_byte': fl / by})
            ctx.prof_enable(False)
        finally:
            ctx.set_lanes(True)
        hbm_rows.sort(key=lambda r: -r['ms'])

    # ---- extra legs (after the timed region, N = 1 only).  A default run keeps exact_fp32 (the exact-fp32-multiply twin of
    # `value`); everything else -- host-resident input, the A/B legs, the CLI from files, a CPU baseline per config -- is behind
    # --extras, so that the driver's run stays well inside a minute
    extras = {}
    leg_s = {}
    t_leg = time.perf_counter()
    if world == 1 and not args.no_extras:
        ctx.set_exact(True)
        try:
            run_step(models, imgs[0], args)
            t, _ = timed_steps(models, imgs, args, dev, args.exact_steps)
        finally:
            ctx.set_exact(False)
        extras['exact_fp32'] = {'value': args.exact_steps / t, 'ms_per_step': 1e3 * t / args.exact_steps,
                                'steps': args.exact_steps, 'unit': 'micrographs/s',
                                'note': 'every convolution on the fp32-MFMA kernels (tpz_ctx_set_exact): exact fp32 '
                                        'multiplies, peak 157.3 TFLOP/s.  Same patch windows as the 2xf16 path (each layer of a '
                                        'denoise patch computes the rectangle the kept centre depends on); the max-pools, the 1x1 '
                                        'projections and the last conv run as layers of their own here (not fused / folded)',
                                'like_for_like_with': 'value'}
        leg_s['exact_fp32'] = time.perf_counter() - t_leg
    if world == 1 and args.extras:
        extras['pcie_inclusive'] = pcie_inclusive(models, host_imgs, args, dev)
        if args.workload != 'denoise':
            # A/B of the patch raster of the 8-wave launches (scoring stage): row-major XCD runs instead of 8 x 4 blocks of tiles
            ctx.set_raster(False)
            try:
                run_step(models, imgs[0], args)
                t, _ = timed_steps(models, imgs, args, dev, 2)
            finally:
                ctx.set_raster(True)
            extras['row_major_raster'] = {'value': 2 / t, 'ms_per_step': 1e3 * t / 2, 'steps': 2, 'unit': 'micrographs/s',
                                          'note': 'tpz_ctx_set_raster(0): each XCD walks a row-major run of tiles (the round-3 order)'}
        if args.workload != 'extract':
            # A/B of the batched patches: the same step with the patches of the denoise stage launched one by one, alternating
            # on the two patch lanes (the round-3 form)
            ctx.set_batch(0)
            try:
                run_step(models, imgs[0], args)
                l0 = ctx.launches()
                t, _ = timed_steps(models, imgs, args, dev, 2)
                l1 = ctx.launches()
            finally:
                ctx.set_batch(8)
            extras['patch_lanes_unbatched'] = {
                'value': 2 / t, 'ms_per_step': 1e3 * t / 2, 'steps': 2, 'unit': 'micrographs/s', 'launches_per_step': (l1 - l0) / 2,
                'note': 'tpz_ctx_set_batch(0): every patch of the denoise stage launches its own layers, patches alternating on the '
                        'two patch lanes; `value` above issues the same layer of 8 patches as one grid and alternates the BATCHES '
                        'on the lanes -- bit-identical output (tests/test_gpu_denoise.py::test_batched_patches_are_bit_identical; '
                        'same-process A/B of the denoise stage alone: profiles/r04_batch_lanes_ab.txt)'}
            # A/B of the patch windows: the same step with every tensor of every denoise patch computed in full
            ctx.set_roi(False)
            try:
                run_step(models, imgs[0], args)
                t, _ = timed_steps(models, imgs, args, dev, 2)
            finally:
                ctx.set_roi(True)
            extras['full_patch_tensors'] = {
                'value': 2 / t, 'ms_per_step': 1e3 * t / 2, 'steps': 2, 'unit': 'micrographs/s',
                'note': 'tpz_ctx_set_roi(0): every layer of a denoise patch computes its whole 2024^2 tensor, as the '
                        'reference does, although only the 1024^2 centre of a patch is kept; `value` above computes, per '
                        'layer, only the rectangle those kept pixels depend on -- bit-identical output '
                        '(tests/test_gpu_denoise.py::test_patch_windows_are_bit_identical)'}

    # images the 2xf16 path sent back to the fp32 kernels (an activation beyond the f16 range): 0 on this workload
    fp32_reruns = 0
    for key in ('denoise', 'score'):
        if key in models:
            dm = models[key][0].model.device_model if key == 'denoise' else models[key][0].device_model
            fp32_reruns += int(dm.split_stats()[2])
    configs = None
    if rank == 0 and world == 1 and not args.no_configs and args.workload == 'pipeline':
        t_leg = time.perf_counter()
        configs = baseline_configs(ctx, models, imgs, args, dev, with_cpu=args.extras and not args.no_cpu_baseline)
        leg_s['configs'] = time.perf_counter() - t_leg
        if energy:
            # the energy view per config: joules per unit at the config's own board power against what its executed f16 MFMA FLOP
            # cost at the probe's pJ / FLOP (every layer of these configs is on the 2xf16 path: 3 MFMA FLOP per executed FLOP)
            for c in configs.values():
                if isinstance(c, dict) and c.get('power_w_mean') and c.get('executed_tflop'):
                    j = c['power_w_mean'] * c['ms'] * 1e-3
                    fl = 3.0 * c['executed_tflop'] * energy['probe_pj_per_f16_flop']
                    c['joules'], c['mfma_floor_joules'], c['energy_frac'] = j, fl, fl / j if j > 0 else None
        if args.extras:
            t_leg = time.perf_counter()
            try:
                extras['cli_inclusive'] = cli_inclusive(args, dev)
            except Exception as e:                               # (a leg of its own: never takes the line down)
                extras['cli_inclusive'] = {'error': f'{type(e).__name__}: {e}'}
            leg_s['cli_inclusive'] = time.perf_counter() - t_leg

    if rank == 0:
        out = {
            'metric': 'micrographs/sec (4096x4096 fp32) denoise+score, NMS parity',
            'value': total_steps / dt,
            'unit': 'micrographs/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': 1e3 * dt / max(1, args.steps),
            'higher_is_better': True,
            'scaling': args.scaling,
            'rccl_world': rccl_world,
            'host_placement': {'rank0_cpus': len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else None,
                               'pinned_to_gpu_numa_node': bool(os.environ.get('TOPAZ_AMD_RANK_CPUS')),
                               'host_cpus_before_pinning': len(_ORIG_AFFINITY) if _ORIG_AFFINITY is not None else None},
            'rank_ms_per_step': {'min': min(rank_ms), 'max': max(rank_ms), 'all': [round(v, 3) for v in rank_ms]},
            # host CPU time each rank process spent per step (launch thread, runtime, Python): what 8 ranks on one host add up to
            'rank_host_cpu_ms_per_step': {'max': max(rank_cpu_ms), 'all': [round(v, 3) for v in rank_cpu_ms]},
            'vs_baseline': None,
            'dtype': 'f32 (convs: fp32 operands carried as two f16 halves on the f16 MFMA, three exact products per '
                     'multiply-add accumulated in f32 -- fp32-level error, fp32-MFMA re-run on f16-range overflow; '
                     'NMS in fp32; the exact-fp32-multiply rate is reported as exact_fp32)',
            'data': 'synthetic (N(0,1) micrographs seed 1000+i; seeded random weights of the named architectures)',
            'config': {
                'workload': {'pipeline': 'denoise(unet b11/t5 nf48, -s 1024 -p 500) -> score(resnet8 u64, filled) -> NMS(r=14,t=-6)',
                             'extract': 'score(resnet8 u64, filled) -> NMS(r=14,t=-6)',
                             'denoise': 'denoise(unet b11/t5 nf48, -s 1024 -p 500)'}[args.workload],
                'image': f'{args.size}x{args.size} fp32', 'images_per_rank': args.steps, 'images_total': total_steps,
                'parallelism': f'one micrograph per rank x{world}; RCCL gather of pick tables',
                'picks_per_image': n_picks / max(1, len(scs)) if have_picks else None,
                'patch_windows': 'on: a denoise patch computes, layer by layer, only what its kept 1024^2 centre depends on '
                                 '(bit-identical to computing the 2024^2 tensors in full; full_patch_tensors = the A/B leg)',
            },
            'gather_ms': 1e3 * t_gather,
            'launches_per_step': launches_per_step,
            'fp32_reruns': fp32_reruns,
            'gpu_clock_power': sampler.summary(),
            'energy': energy,
            # the dominant kernel = the conv instantiation with the most time in a step (live HIP-event timing of the
            # timed steps' own launches); `achieved` is algorithmic (fp32-equivalent) FLOP/s
            'roofline': {
                'bound': 'mfma', 'kernel': dom_name,
                'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s', 'frac': achieved / peak, 'traffic': traffic,
                'traffic_detail': traffic_detail,
                'peak_basis': ('f16 dense MFMA peak 2500 TFLOP/s / 3 MFMAs per fp32-equivalent MAC (2xf16 split)'
                               if is_split else 'fp32 MFMA peak'),
                'sustained_mfma': sustained,
                'launches_per_step': dom_n, 'avg_launch_ms': dom_ms / max(1e-9, dom_n),
                'timed_over': ('the timed steps themselves (HIP events on the launch stream)' if not args.no_kernel_timing
                               else 'one extra step after the timed region'),
                'algorithmic_tflop_per_launch': dom_flops / max(1e-9, dom_n) / 1e12,
                'share_of_step': dom_ms / (1e3 * dt / args.steps) if dt > 0 else None,
                'class_conv_mfma_fp32': cls_f32, 'class_conv_split_2xf16': cls_split,
                'conv_kernel_ms_per_step': conv_ms, 'conv_algorithmic_tflop_per_step': conv_flops / 1e12,
                'top_kernels': [{'kernel': k[0], 'ms': k[1], 'launches': k[2], 'tflops': k[3] / k[1] / 1e9}
                                for k in kernels_iso[:6]],
                'class_and_top_kernels_from': 'one extra step with the patch lanes off (kernels timed in isolation)',
                **({} if not args.no_kernel_timing else other),
                'hbm_bound_kernels': {'peak': HBM_PEAK_BYTES / 1e9, 'unit': 'GB/s', 'rows': hbm_rows[:6],
                                      'from': 'one extra step, every launch timed, patch lanes off; bytes = inputs (with halo) + '
                                              'weights read once, outputs written once'},
                'coverage': ('convolution launches of >= 20 GFLOP (the rest, elementwise and NMS kernels are not timed inside '
                             'the timed region)' if not args.no_kernel_timing else 'every launch of one extra step'),
            },
            **extras,
        }
        if step_t:
            n = len(step_t)
            w = min(32, n // 2) or 1
            d_ms = [1e3 * (b - a) for a, b in zip([t0] + step_t[:-1], step_t)]
            ks = sorted(mem_used)
            out['soak'] = {
                'images': n, 'first_ms_per_step': sum(d_ms[:w]) / w, 'last_ms_per_step': sum(d_ms[-w:]) / w, 'window': w,
                'slowest_step_ms': max(d_ms), 'fastest_step_ms': min(d_ms),
                'device_bytes_in_use_after_step': {str(k + 1): int(mem_used[k]) for k in ks},
                'pool_growth_bytes_after_warm_up': int(mem_used[ks[-1]] - mem_used[ks[0]]) if len(ks) >= 2 else 0,
                'peak_device_gb': max(mem_used.values()) / 1e9 if mem_used else None,
                'fp32_reruns': fp32_reruns,
                'what': f'--scaling strong: the fixed job of {args.images} micrographs ({n} on this rank, cycling through '
                        f'{n_res} resident inputs) through denoise -> score -> NMS; per-step host clock (each step ends with the '
                        'host read of its pick count), device memory from hipMemGetInfo'}
        if configs is not None:
            out['configs'] = configs
        if world == 1 and not args.no_cpu_baseline:
            t_leg = time.perf_counter()
            keep = {}
            out['cpu_baseline'] = cpu_baseline(models, args, keep)
            leg_s['cpu_baseline'] = time.perf_counter() - t_leg
            # the same arrays through the HIP path: parity as part of the line (and of the exit status)
            out['parity'] = parity_vs_oracle(models, keep, args, dev)
        out['leg_seconds'] = {k: round(v, 2) for k, v in leg_s.items()}
        print(json.dumps(out))
        if 'parity' in out and not out['parity']['ok']:
            sys.stderr.write(f"bench.py: PARITY FAILED against the oracle: {out['parity']}\n")
            sys.exit(3)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
