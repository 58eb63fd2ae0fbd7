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
                volume_num: int = 1, total_volumes: int = 1, verbose: bool = True) -> np.ndarray:
        x = self._to_device(tomo)
        y = self.model.device_model.denoise_3d(x, patch_size, padding)
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


# ---- file-level drivers (denoise.py:419-557) ----------------------------------------------------------
def denoise_stack(path: str, output_path: str, models: List[Denoise], lowpass: float = 1, pixel_cutoff: float = 0,
                  gaus=None, inv_gaus=None, deconvolve: bool = True, deconv_patch: int = 1, patch_size: int = 1024,
                  padding: int = 500, normalize: bool = True, use_cuda: bool = True):
    from . import mrc
    with open(path, 'rb') as f:
        content = f.read()
    stack, header, extended_header = mrc.parse(content)
    print('# denoising stack with shape:', stack.shape, file=sys.stderr)
    denoised = np.zeros_like(stack)
    for i in range(len(stack)):
        denoised[i] = denoise_image(stack[i], models, lowpass=lowpass, cutoff=pixel_cutoff, gaus=gaus, inv_gaus=inv_gaus,
                                    deconvolve=deconvolve, deconv_patch=deconv_patch, patch_size=patch_size,
                                    padding=padding, normalize=normalize, use_cuda=use_cuda)
        print('# {} of {} completed.'.format(i + 1, len(stack)), file=sys.stderr, end='\r')
    print('', file=sys.stderr)
    print('# writing to', output_path, file=sys.stderr)
    with open(output_path, 'wb') as f:
        mrc.write(f, denoised, header=header, extended_header=extended_header)
    return denoised


def denoise_stream(micrographs: List[str], output_path: str, format: str = 'mrc', suffix: str = '',
                   models: List[Denoise] = None, lowpass: float = 1, pixel_cutoff: float = 0, gaus=None, inv_gaus=None,
                   deconvolve: bool = True, deconv_patch: int = 1, patch_size: int = 1024, padding: int = 500,
                   normalize: bool = True, use_cuda: bool = True):
    """With WORLD_SIZE > 1 (torchrun) rank r denoises micrographs r, r+world, ... and writes its own files."""
    from . import parallel
    from .utils.image import load_image, save_image
    rank, _, world = parallel.init_from_env()
    total = len(micrographs)
    denoised = []
    if output_path is not None and output_path != '':
        os.makedirs(output_path, exist_ok=True)
    for count, idx in enumerate(parallel.shard_indices(total, rank, world)):
        path = micrographs[idx]
        name, _ = os.path.splitext(os.path.basename(path))
        image = load_image(path, make_image=False)
        image, header, extended_header = image if type(image) is tuple else (image, None, None)
        mic = denoise_image(image, models, lowpass=lowpass, cutoff=pixel_cutoff, gaus=gaus, inv_gaus=inv_gaus,
                            deconvolve=deconvolve, deconv_patch=deconv_patch, patch_size=patch_size, padding=padding,
                            normalize=normalize, use_cuda=use_cuda)
        denoised.append(mic)
        if not output_path:
            if suffix == '' or suffix is None:
                suffix = '.denoised'
            no_ext, ext = os.path.splitext(path)
            outpath = no_ext + suffix + '.' + format
        else:
            outpath = output_path + os.sep + name + suffix + '.' + format
        save_image(mic, outpath, header=header, extended_header=extended_header)
        print(f'# {count + 1} of {total} completed.', file=sys.stderr, end='\r')
    print('', file=sys.stderr)
    return denoised


def denoise_tomogram(path: str, model: Denoise3D, outdir: str = None, suffix: str = '', patch_size: int = 96,
                     padding: int = 48, volume_num: int = 1, total_volumes: int = 1, gaus=None, verbose: bool = True):
    from . import mrc
    name = os.path.basename(path)
    with open(path, 'rb') as f:
        content = f.read()
    tomo, header, extended_header = mrc.parse(content)
    tomo = tomo.astype(np.float32)
    denoised = model.denoise(tomo, patch_size=patch_size, padding=padding, batch_size=1, volume_num=volume_num,
                             total_volumes=total_volumes, verbose=verbose)
    # (the reference filters `tomo`, not `denoised`, and writes `denoised`: denoise.py:509,528-529)
    tomo = gaus.apply(tomo) if gaus is not None else tomo
    if not outdir:
        if suffix == '':
            suffix = '.denoised'
        no_ext, ext = os.path.splitext(path)
        outpath = no_ext + suffix + ext
    else:
        no_ext, ext = os.path.splitext(name)
        outpath = outdir + os.sep + no_ext + suffix + ext
    header = header._replace(mode=2, amin=denoised.min(), amax=denoised.max(), amean=denoised.mean())
    with open(outpath, 'wb') as f:
        mrc.write(f, denoised, header=header, extended_header=extended_header)
    return tomo


def denoise_tomogram_stream(volumes: List[str], model: Denoise3D, output_path: str, suffix: str = '', gaus: float = None,
                            patch_size: int = 96, padding: int = 48, verbose: bool = True, use_cuda: bool = True):
    from . import parallel
    rank, _, world = parallel.init_from_env()
    total = len(volumes)
    denoised = []
    if output_path is not None and output_path != '':
        os.makedirs(output_path, exist_ok=True)
    if gaus is not None and gaus > 0:
        raise NotImplementedError('3-D Gaussian post-filter: the reference applies it to the input and discards it '
                                  '(denoise.py:509); not implemented')
    for count, idx in enumerate(parallel.shard_indices(total, rank, world)):
        volume = denoise_tomogram(volumes[idx], model, outdir=output_path, suffix=suffix, patch_size=patch_size,
                                  padding=padding, volume_num=idx + 1, total_volumes=total, gaus=None, verbose=verbose)
        denoised.append(volume)
        print(f'# {count + 1} of {total} tomograms denoised.', file=sys.stderr, end='\r')
    print('', file=sys.stderr)
    return denoised
