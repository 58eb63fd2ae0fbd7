"""Mirror of the inference side of topaz/denoise.py: Denoise (:245-332), Denoise3D (:336-377),
denoise_image (:382-416), denoise_stack (:419-447), denoise_stream (:450-490), denoise_tomogram
(:495-530) and denoise_tomogram_stream (:533-557).

Patching, per-patch normalisation (torch mean / unbiased std), the network and the stitching all
run on the device inside tpz_denoise_2d / tpz_denoise_3d; only the image goes in and the denoised
image comes out.  The reference's lowpass / deconvolve branches crash in v0.3.18
and raise NotImplementedError here.
"""
from __future__ import annotations

import os
import sys
from typing import List, Optional, Union

import numpy as np
import torch

from . import runtime as rt
from .denoising.models import DenoiseNet, load_model
from .filters import GaussianDenoise, InvGaussianFilter


class Denoise:
    """Object for micrograph denoising utilities (denoise.py:245-332)."""

    def __init__(self, model: Union[DenoiseNet, str], use_cuda: bool = True, dims: int = 2):
        if isinstance(model, DenoiseNet):
            self.model = model
        elif isinstance(model, str):
            try:
                self.model = load_model(model)
            except Exception as e:
                raise ValueError('Unable to load model: ' + model) from e
        else:
            raise TypeError('Unrecognized model:' + str(model))
        self.model.cuda()                       # the MI355X path has no CPU mode
        self.device = self.model.device_model.ctx.torch_device()
        self.dims = dims
        self.use_cuda = True

    def __call__(self, input):
        return self._denoise(input)

    def _to_device(self, x) -> torch.Tensor:
        return rt.as_device_f32(x, self.model.device_model.ctx)

    def _denoise(self, input) -> np.ndarray:
        """normalise by the array's own mean / unbiased std, model, un-normalise (denoise.py:274-296)"""
        x = self._to_device(input)
        if x.dim() == self.dims + 1 and x.shape[0] == 1:
            x = x[0]
        if x.dim() != self.dims:
            raise ValueError(f'expected a {self.dims}-D array, got shape {tuple(x.shape)}')
        dm = self.model.device_model
        y = dm.denoise_3d(x, -1, 0) if self.dims == 3 else dm.denoise_2d(x, -1, 0)
        return y.cpu().numpy()

    def denoise_patches(self, x, patch_size: int, padding: int = 128) -> np.ndarray:
        return self._run2d(x, patch_size, padding, force_patches=True)

    def denoise(self, x, patch_size: int = -1, padding: int = 128) -> np.ndarray:
        return self._run2d(x, patch_size, padding)

    def denoise_device(self, x: torch.Tensor, patch_size: int = -1, padding: int = 128) -> torch.Tensor:
        """same as denoise() but device tensor in, device tensor out (no PCIe traffic)"""
        if self.dims == 3:
            return self.model.device_model.denoise_3d(x, -1, 0)
        return self.model.device_model.denoise_2d(x, patch_size, padding)

    def _run2d(self, x, patch_size, padding, force_patches=False) -> np.ndarray:
        if self.dims == 3:
            return self._denoise(x)
        xd = self._to_device(x)
        if force_patches and not (patch_size > 0 and (patch_size + padding < xd.shape[0] or patch_size + padding < xd.shape[1])):
            # denoise_patches called directly on a small image still tiles it (denoise.py:307-322)
            out = np.zeros(tuple(xd.shape), dtype=np.float32)
            H, W = xd.shape
            for i in range(0, H, patch_size):
                for j in range(0, W, patch_size):
                    si, ei = max(0, i - padding), min(H, i + patch_size + padding)
                    sj, ej = max(0, j - padding), min(W, j + patch_size + padding)
                    yij = self.model.device_model.denoise_2d(xd[si:ei, sj:ej].contiguous(), -1, 0).cpu().numpy()
                    oi, oj = i - si, j - sj
                    out[i:i + patch_size, j:j + patch_size] = yij[oi:oi + patch_size, oj:oj + patch_size]
            return out
        return self.model.device_model.denoise_2d(xd, patch_size, padding).cpu().numpy()


class Denoise3D(Denoise):
    """Object for denoising tomograms (denoise.py:336-377)."""

    def __init__(self, model, use_cuda: bool = True, dims: int = 3):
        super().__init__(model, use_cuda, dims=3)

    def denoise(self, tomo: np.ndarray, patch_size: int = 96, padding: int = 48, batch_size: int = 1,
                volume_num: int = 1, total_volumes: int = 1, verbose: bool = True, across_ranks: bool = False) -> np.ndarray:
        """across_ranks=True (multi-process launch, every rank holding the same tomogram): the ceil(n / patch)^3 tiles are
        dealt round-robin to the ranks -- they are independent (datasets.py:412-468), the global mean / std every rank
        computes over the whole volume is the same deterministic reduction -- and the ranks' partial volumes are summed onto
        rank 0 with one RCCL reduce; the other ranks return None."""
        x = self._to_device(tomo)
        dm = self.model.device_model
        if across_ranks and patch_size >= 1:
            from . import parallel
            rank, _, world = parallel.init_from_env()
            part = dm.denoise_3d(x, patch_size, padding, shard=rank, n_shards=world)
            if parallel.collective_device(0).type == 'cpu':       # gloo rehearsal (ranks sharing a GPU): reduce host tensors
                part = part.cpu()
            y = parallel.sum_to_root(part)
            if y is None:
                return None
        else:
            y = dm.denoise_3d(x, patch_size, padding)
        if verbose:
            print(f'# [{volume_num}/{total_volumes}] 100%', file=sys.stderr, end='\r')
            print(' ' * 100, file=sys.stderr, end='\r')
        return y.cpu().numpy().astype(np.asarray(tomo).dtype, copy=False)


def denoise_image(mic: np.ndarray, models: List[Denoise], lowpass=1, cutoff=0, gaus: GaussianDenoise = None,
                  inv_gaus: InvGaussianFilter = None, deconvolve=False, deconv_patch=1, patch_size=-1, padding=0,
                  normalize=False, use_cuda=True) -> np.ndarray:
    """denoise_image (denoise.py:382-416).  numpy mean / POPULATION std here (unlike _denoise)."""
    if lowpass > 1:
        raise NotImplementedError('lowpass: crashes in the reference (denoise.py:386 shadows the function)')
    mic = np.asarray(mic)
    mu, std = mic.mean(), mic.std()
    x = (mic - mu) / std
    if cutoff > 0:
        x[(x < -cutoff) | (x > cutoff)] = 0
    if gaus is not None:
        x = gaus.apply(x)
    elif inv_gaus is not None:
        x = inv_gaus.apply(x)
    elif deconvolve:
        raise NotImplementedError('deconvolve: crashes in the reference (denoise.py:404 on ndarray input)')
    if len(models) == 0:
        # `-m none` (commands/denoise.py:100-106 still builds a Denoise around no model): the pre-filtered image passes through
        out = np.asarray(x, dtype=np.float32)
    else:
        out = sum(model.denoise(x, patch_size=patch_size, padding=padding) for model in models) / len(models)
    if normalize:
        out = (out - out.mean()) / out.std()
    else:
        out = std * out + mu
    return out


def denoise_image_device(x: torch.Tensor, models: List[Denoise], patch_size: int = -1, padding: int = 0,
                         normalize: bool = False) -> torch.Tensor:
    """denoise_image (denoise.py:382-416) for the plain case -- no pixel cutoff, no Gaussian / inverse filter -- with the
    micrograph staying on the device: population mean / std (the reference's numpy statistics, here a deterministic fp64
    reduction on the GPU), (x - mu) / std, the networks' average, then either re-normalisation or std * y + mu.  The CLI path
    spends no host pass over the 16.7 M pixels this way (the numpy version costs ~130 ms per 4096^2 micrograph, 7x the GPU
    work of the network)."""
    from . import runtime as rt
    mu, std = rt.mean_std(x, unbiased=False)
    xn = rt.normalize(x, mu, std)              # (x - mu) / std as numpy rounds it; std == 0 -> inf / nan like upstream
    out = None
    for model in models:
        y = model.denoise_device(xn, patch_size, padding)
        out = y if out is None else out + y
    if len(models) > 1:
        out = out / len(models)
    if normalize:
        m2, s2 = rt.mean_std(out, unbiased=False)
        return rt.normalize(out, m2, s2)
    return rt.affine(out, std, mu)


# ---- file-level drivers -----------------------------------------------------------------------------------------------
# Counterparts of denoise_stack (denoise.py:419-447), denoise_stream (:450-490), denoise_tomogram (:495-530) and
# denoise_tomogram_stream (:533-557): same arguments, same files on disk.  Structure here: a Job names one input and its
# output path; `_run_jobs` overlaps the disk read of job i+1 and the write of job i-1 with the GPU work of job i (one
# reader and one writer thread around the device loop) and shards the jobs over the ranks when launched multi-process.
class _Job:
    __slots__ = ('index', 'src', 'dst')

    def __init__(self, index: int, src: str, dst: str):
        self.index, self.src, self.dst = index, src, dst


def _output_path(src: str, outdir: Optional[str], suffix: str, ext: str) -> str:
    """<outdir>/<name><suffix><ext>, or next to the input with '.denoised' when no directory is given
    (denoise.py:476-483, 512-520)"""
    stem, _ = os.path.splitext(src)
    if not outdir:
        return stem + (suffix or '.denoised') + ext
    return os.path.join(outdir, os.path.basename(stem) + suffix + ext)


def _run_jobs(jobs: List[_Job], read, process, write, progress) -> list:
    """read(job) -> item; process(job, item) -> result; write(job, item, result).  Reads run one job ahead and writes one
    job behind the processing, each on its own thread; exceptions of either surface here."""
    from concurrent.futures import ThreadPoolExecutor
    results = []
    if not jobs:
        return results
    with ThreadPoolExecutor(1, 'tpz-read') as reader, ThreadPoolExecutor(1, 'tpz-write') as writer:
        nxt = reader.submit(read, jobs[0])
        pending = None
        for n, job in enumerate(jobs):
            item = nxt.result()
            if n + 1 < len(jobs):
                nxt = reader.submit(read, jobs[n + 1])
            result = process(job, item)
            if pending is not None:
                pending.result()
            pending = writer.submit(write, job, item, result)
            results.append(result)
            progress(n + 1)
        if pending is not None:
            pending.result()
    return results


def denoise_stack(path: str, output_path: str, models: List[Denoise], lowpass: float = 1, pixel_cutoff: float = 0,
                  gaus=None, inv_gaus=None, deconvolve: bool = True, deconv_patch: int = 1, patch_size: int = 1024,
                  padding: int = 500, normalize: bool = True, use_cuda: bool = True):
    """every section of one MRC stack, written back as one stack with the input's header"""
    from . import mrc
    with open(path, 'rb') as f:
        stack, header, extended_header = mrc.parse(f.read())
    print('# denoising stack with shape:', stack.shape, file=sys.stderr)
    out = np.zeros_like(stack)
    for k, section in enumerate(stack):
        out[k] = denoise_image(section, models, lowpass=lowpass, cutoff=pixel_cutoff, gaus=gaus, inv_gaus=inv_gaus,
                               deconvolve=deconvolve, deconv_patch=deconv_patch, patch_size=patch_size, padding=padding,
                               normalize=normalize, use_cuda=use_cuda)
        print(f'# {k + 1} of {len(stack)} completed.', file=sys.stderr, end='\r')
    print('', file=sys.stderr)
    print('# writing to', output_path, file=sys.stderr)
    with open(output_path, 'wb') as f:
        mrc.write(f, out, header=header, extended_header=extended_header)
    return out


def denoise_stream(micrographs: List[str], output_path: str, format: str = 'mrc', suffix: str = '',
                   models: List[Denoise] = None, lowpass: float = 1, pixel_cutoff: float = 0, gaus=None, inv_gaus=None,
                   deconvolve: bool = True, deconv_patch: int = 1, patch_size: int = 1024, padding: int = 500,
                   normalize: bool = True, use_cuda: bool = True, return_images: bool = True):
    """one output image per micrograph; rank r of a multi-process launch takes micrographs r, r + world, ...
    Returns the denoised micrographs like the reference (denoise.py:450-490 keeps every one of them in memory);
    return_images=False (the CLI, which ignores them) returns the output paths instead."""
    from . import parallel
    from .utils.image import load_image, save_image
    rank, _, world = parallel.init_from_env()
    if output_path:
        os.makedirs(output_path, exist_ok=True)
    jobs = [_Job(i, micrographs[i], _output_path(micrographs[i], output_path, suffix, '.' + format))
            for i in parallel.shard_indices(len(micrographs), rank, world)]

    plain = (models and lowpass <= 1 and pixel_cutoff <= 0 and gaus is None and inv_gaus is None and not deconvolve and
             all(m.dims == 2 for m in models))
    if plain:
        return _denoise_stream_device(jobs, models, patch_size, padding, normalize, len(micrographs), return_images)

    def read(job):
        loaded = load_image(job.src, make_image=False)
        return loaded if isinstance(loaded, tuple) else (loaded, None, None)

    def process(job, item):
        return denoise_image(item[0], models, lowpass=lowpass, cutoff=pixel_cutoff, gaus=gaus, inv_gaus=inv_gaus,
                             deconvolve=deconvolve, deconv_patch=deconv_patch, patch_size=patch_size, padding=padding,
                             normalize=normalize, use_cuda=use_cuda)

    def write(job, item, mic):
        save_image(mic, job.dst, header=item[1], extended_header=item[2])

    total = len(micrographs)
    out = _run_jobs(jobs, read, process, write, lambda n: print(f'# {n} of {total} completed.', file=sys.stderr, end='\r'))
    print('', file=sys.stderr)
    return out


def _denoise_stream_device(jobs: List[_Job], models: List[Denoise], patch_size: int, padding: int, normalize: bool, total: int,
                           return_images: bool = False) -> list:
    """the plain `topaz denoise` loop with every pass over the pixels on the device: a reader thread decodes micrograph i+1
    into pinned memory and queues its upload (extract.ImageFeed), the GPU denoises micrograph i (denoise_image_device), its
    result is copied into a pinned slot by the copy stream and a writer thread writes micrograph i-1 from there."""
    import queue
    import threading
    from . import runtime as rt
    from .extract import ImageFeed
    from .utils.image import save_image
    results = []
    if not jobs:
        return results
    ctx = models[0].model.device_model.ctx
    by_src = {}
    for job in jobs:
        by_src.setdefault(job.src, []).append(job)
    out_stage, out_bytes, DEPTH = None, 0, 2
    free_slots: 'queue.Queue' = queue.Queue()
    todo: 'queue.Queue' = queue.Queue()
    failed: list = []

    def writer():
        while True:
            item = todo.get()
            if item is None:
                return
            stage, k, shape, dst, header, extended, index, _device_result = item      # (the tensor lives until its copy is done)
            try:
                stage.wait(k)                                # the D2H of this slot has landed
                save_image(stage.host_array(k, shape), dst, header=header, extended_header=extended)
                if return_images:
                    results[index] = np.array(stage.host_array(k, shape), copy=True)
            except BaseException as e:                       # surfaces in the main thread
                failed.append(e)
            finally:
                free_slots.put((stage, k))

    th = threading.Thread(target=writer, daemon=True)
    th.start()
    retired = []
    try:
        for n, (path, x, header, extended) in enumerate(ImageFeed([j.src for j in jobs], ctx, headers=True)):
            job = by_src[path].pop(0)
            y = denoise_image_device(x, models, patch_size, padding, normalize)
            nbytes = y.numel() * 4
            if out_stage is None or nbytes > out_bytes:
                # (a larger image: a new ring; the old one is closed once the writer has drained it)
                for _ in range(DEPTH if out_stage is not None else 0):
                    free_slots.get()
                if out_stage is not None:
                    retired.append(out_stage)
                out_bytes = (nbytes + (1 << 20) - 1) & ~((1 << 20) - 1)
                out_stage = rt.Stage(ctx, out_bytes, DEPTH)
                for k in range(DEPTH):
                    free_slots.put((out_stage, k))
            stage, k = free_slots.get()
            if failed:
                raise failed[0]
            stage.download(k, y)
            results.append(job.dst)
            todo.put((stage, k, tuple(y.shape), job.dst, header, extended, n, y))
            print(f'# {n + 1} of {total} completed.', file=sys.stderr, end='\r')
    finally:
        todo.put(None)
        th.join()
        for st in retired + ([out_stage] if out_stage is not None else []):
            st.close()
    if failed:
        raise failed[0]
    print('', file=sys.stderr)
    return results


def _tomogram_io():
    from . import mrc

    def read(job):
        with open(job.src, 'rb') as f:
            tomo, header, extended = mrc.parse(f.read())
        return tomo.astype(np.float32), header, extended

    def write(job, item, denoised):
        # 3-D denoise refreshes the density statistics of the header (denoise.py:523-526)
        header = item[1]._replace(mode=2, amin=denoised.min(), amax=denoised.max(), amean=denoised.mean())
        with open(job.dst, 'wb') as f:
            mrc.write(f, denoised, header=header, extended_header=item[2])
    return read, write


def denoise_tomogram(path: str, model: Denoise3D, outdir: str = None, suffix: str = '', patch_size: int = 96,
                     padding: int = 48, volume_num: int = 1, total_volumes: int = 1, gaus=None, verbose: bool = True):
    """one tomogram; returns the INPUT volume, Gaussian-filtered when `gaus` is given -- what the reference returns
    (it filters `tomo`, writes `denoised`: denoise.py:509,528-530)"""
    read, write = _tomogram_io()
    job = _Job(0, path, _output_path(path, outdir, suffix, os.path.splitext(path)[1]))
    item = read(job)
    denoised = model.denoise(item[0], patch_size=patch_size, padding=padding, batch_size=1, volume_num=volume_num,
                             total_volumes=total_volumes, verbose=verbose)
    write(job, item, denoised)
    return gaus.apply(item[0]) if gaus is not None else item[0]


def denoise_tomogram_stream(volumes: List[str], model: Denoise3D, output_path: str, suffix: str = '', gaus: float = None,
                            patch_size: int = 96, padding: int = 48, verbose: bool = True, use_cuda: bool = True):
    """tomograms (or, when there are fewer tomograms than ranks, their tiles) sharded over the ranks of a multi-process
    launch; the next volume is read while this one is denoised"""
    from . import parallel
    if gaus is not None and gaus > 0:
        # upstream builds a TWO-dimensional GaussianDenoise here (denoise.py:546, dims defaults to 2) and applies it to
        # the input volume before anything is written (:509): Conv2d rejects the 5-D tensor, so `--gaussian` > 0 has
        # only ever raised.  Same outcome, said plainly.  (denoise_tomogram(gaus=GaussianDenoise(s, dims=3)) works.)
        raise RuntimeError('denoise3d --gaussian: the 2-D Gaussian module cannot filter a volume (upstream fails in '
                           'Conv2d at this point); use 0')
    rank, _, world = parallel.init_from_env()
    if output_path:
        os.makedirs(output_path, exist_ok=True)
    read, write = _tomogram_io()
    total = len(volumes)
    # fewer tomograms than ranks: split each tomogram's TILES over all ranks instead of leaving ranks idle (a 512x512x256
    # volume is 108 independent 192^3 tiles); otherwise whole tomograms are dealt to the ranks
    by_tiles = world > 1 and total < world and patch_size >= 1
    mine = range(total) if by_tiles else parallel.shard_indices(total, rank, world)
    jobs = [_Job(i, volumes[i], _output_path(volumes[i], output_path, suffix, os.path.splitext(volumes[i])[1])) for i in mine]
    inputs = []          # upstream returns the list of (unfiltered) input volumes; kept for the same return value

    def process(job, item):
        inputs.append(item[0])
        return model.denoise(item[0], patch_size=patch_size, padding=padding, batch_size=1, volume_num=job.index + 1,
                             total_volumes=total, verbose=verbose and rank == 0, across_ranks=by_tiles)

    if by_tiles:
        plain_write = write

        def write(job, item, denoised):            # only rank 0 holds the assembled volume
            if denoised is not None:
                plain_write(job, item, denoised)

    _run_jobs(jobs, read, process, write, lambda n: print(f'# {n} of {total} tomograms denoised.', file=sys.stderr, end='\r'))
    print('', file=sys.stderr)
    return inputs
