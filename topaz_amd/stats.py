"""GMM / affine image normalisation on the MI355X: the host side of topaz/stats.py:17-84,277-352
(`pixels_given_radius`, `calculate_pi`, `normalize`, `norm_fit`, `Normalize`, `normalize_images`).

The mixture fit itself (`gmm_fit`, stats.py:120-203: twelve initialisations x up to `num_iters` EM iterations over
the sampled pixels) runs on the device through `tpz_gmm_fit` -- one fused E-step pass per iteration, statistics in
fp64; the quantile splits and the optional random sub-sample are taken on the host exactly as the reference takes
them (`np.quantile`, `np.random.choice` on the global NumPy RNG), so a seeded run fits the same pixels."""
from __future__ import annotations

import json
import os
import sys
from typing import List

import numpy as np

INIT_PIS = (0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 0.95, 0.98, 1.0)     # stats.py:91
_FIT_ARRAYS = ('mus', 'stds', 'pis', 'logps')                                    # per-initialisation results (metadata)


def _lattice_points(radius: int, dims: int) -> int:
    """integer offsets within `radius` of the origin: a disc (dims 2) or a ball (dims 3)"""
    k = np.arange(-int(radius), int(radius) + 1, dtype=np.int64) ** 2
    d2 = k[:, None] + k[None, :]
    if dims == 3:
        d2 = d2[:, :, None] + k[None, None, :]
    return int(np.count_nonzero(d2 <= int(radius) ** 2))


def pixels_given_radius(radius, dims=2):
    """Pixels the reference attributes to a particle of that radius (stats.py:17-26).  Its mask always lives on a CUBE of side
    2r + 1 whatever `dims` is; in 2-D the z axis simply does not enter the distance, so the disc is counted once per z plane --
    2r + 1 times.  `calculate_pi` feeds that number on, so it is kept."""
    if dims == 3:
        return _lattice_points(radius, 3)
    return _lattice_points(radius, 2) * (2 * int(radius) + 1)


def calculate_pi(expected_num_particles, radius, total_pixels, dims=2):
    """prior fraction of particle pixels (stats.py:29-34)"""
    return pixels_given_radius(radius, dims=dims) * expected_num_particles / total_pixels


def norm_fit(x, alpha=900, beta=1, scale=1, num_iters=100, use_cuda=True, verbose=False, tol=1e-3):
    """stats.py:87-117: fit every initialisation, keep the one with the largest log-probability.
    Returns (mu, std, pi, logp, mus, stds, pis, logps)."""
    from . import runtime as rt
    x = np.ascontiguousarray(np.asarray(x, dtype=np.float32).ravel())
    pis = np.array(INIT_PIS, dtype=np.float64)
    splits = np.quantile(x, 1 - pis).astype(np.float64)
    mus, stds, pis_fit, logps = rt.gmm_fit(x, pis, splits, alpha=alpha, beta=beta, scale=scale, num_iters=num_iters,
                                           tol=tol)
    i = int(np.argmax(logps))
    return mus[i], stds[i], pis_fit[i], logps[i], mus, stds, pis_fit, logps


def _fit_pixels(x: np.ndarray, sample: int):
    """the pixels the mixture is fitted on and the weight each one carries: all of them, or every `sample`-th of them drawn
    without replacement from the GLOBAL NumPy RNG -- the reference's contract (stats.py:55-60): a run seeded with
    np.random.seed fits the same pixels here and there"""
    if sample <= 1:
        return x, 1
    n = int(np.round(x.size / sample))
    return np.random.choice(x.ravel(), size=n, replace=False), x.size / n


def normalize(x, alpha=900, beta=1, num_iters=100, sample=1, method='gmm', use_cuda=True, verbose=False):
    """(x - mu) / std as float32 plus the metadata dict of stats.py:37-84; mu, std = the image's own mean and standard
    deviation (`affine`) or the background component of the fitted two-component mixture (`gmm`)."""
    if method == 'affine':
        meta = {'mu': float(x.mean()), 'std': float(x.std()), 'pi': 1}
    else:
        pixels, weight = _fit_pixels(x, sample)
        fit = norm_fit(pixels, alpha=alpha, beta=beta, scale=weight, num_iters=num_iters, verbose=verbose)
        meta = dict(zip(('mu', 'std', 'pi', 'logp') + _FIT_ARRAYS, fit))
        meta.update(alpha=alpha, beta=beta, sample=sample)
    return ((x - meta['mu']) / meta['std']).astype(np.float32), meta


def _jsonable(meta: dict) -> dict:
    """the metadata with NumPy arrays and scalars as plain lists and floats"""
    out = {}
    for k, v in meta.items():
        if k in _FIT_ARRAYS:
            out[k] = np.asarray(v).tolist()
        elif isinstance(v, np.generic):
            out[k] = v.item()
        else:
            out[k] = v
    return out


class Normalize:
    """per-file worker of `topaz normalize` / `topaz preprocess` (stats.py:277-331): load, optional Fourier down-sampling,
    normalisation, one output file per requested format, optional `<name>.metadata.json`."""

    def __init__(self, dest, scale, affine, num_iters, alpha, beta, sample, metadata, formats, use_cuda=True):
        self.dest, self.scale, self.formats, self.metadata = dest, scale, formats, metadata
        self.fit_args = dict(alpha=alpha, beta=beta, num_iters=num_iters, sample=sample, method='affine' if affine else 'gmm')

    def _load(self, path):
        """(float32 pixels, MRC header or None, extended header or None), down-sampled by `scale` with the header's size
        fields following"""
        from .utils.image import downsample, load_image
        loaded = load_image(path, make_image=False)
        pixels, header, extended = loaded if isinstance(loaded, tuple) else (loaded, None, None)
        pixels = pixels.astype(np.float32)
        if self.scale > 1:
            pixels = downsample(pixels, self.scale)
            if header:
                header = header._replace(ny=pixels.shape[0], nx=pixels.shape[1])
        return pixels, header, extended

    def __call__(self, path):
        from .utils.image import save_image
        pixels, header, extended = self._load(path)
        pixels, meta = normalize(pixels, **self.fit_args)
        name = os.path.splitext(os.path.basename(path))[0]
        stem = os.path.join(self.dest, name)
        for fmt in self.formats:
            save_image(pixels, stem, f=fmt, header=header, extended_header=extended)
        if self.metadata:
            with open(stem + '.metadata.json', 'w') as fh:
                json.dump(_jsonable(meta), fh, indent=4)
        return name


def normalize_images(paths: List[str], dest: str, num_workers: int, scale: int, affine: bool, niters: int, alpha: float,
                     beta: float, sample: int, metadata: bool, formats: List[str], use_cuda: bool = True,
                     verbose: bool = False):
    """stats.py:334-352 (`num_workers` is accepted and unused: the fit runs on the GPU, one image at a time)"""
    os.makedirs(dest, exist_ok=True)
    worker = Normalize(dest, scale, affine, niters, alpha, beta, sample, metadata, formats)
    for path in paths:
        name = worker(path)
        if verbose:
            print('# processed:', name, file=sys.stderr)
