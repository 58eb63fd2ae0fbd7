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


def pixels_given_radius(radius, dims=2):
    """number of pixels of a disc / ball of that radius (stats.py:17-26)"""
    g = np.linspace(-radius, radius, 2 * radius + 1)
    xx, yy, zz = np.meshgrid(g, g, g)
    d2 = xx ** 2 + yy ** 2
    if dims == 3:
        d2 = d2 + zz ** 2
    return int((d2 <= radius ** 2).astype(int).sum())


def calculate_pi(expected_num_particles, radius, total_pixels, dims=2):
    """stats.py:29-34"""
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


def normalize(x, alpha=900, beta=1, num_iters=100, sample=1, method='gmm', use_cuda=True, verbose=False):
    """stats.py:37-84.  Returns (normalised float32 image, metadata dict)."""
    if method == 'affine':
        mu, std = float(x.mean()), float(x.std())
        return ((x - mu) / std).astype(np.float32), {'mu': mu, 'std': std, 'pi': 1}
    x_sample, scale = x, 1
    if sample > 1:
        n = int(np.round(x.size / sample))
        scale = x.size / n
        x_sample = np.random.choice(x.ravel(), size=n, replace=False)
    mu, std, pi, logp, mus, stds, pis, logps = norm_fit(x_sample, alpha=alpha, beta=beta, scale=scale,
                                                        num_iters=num_iters, verbose=verbose)
    out = ((x - mu) / std).astype(np.float32)
    meta = {'mu': mu, 'std': std, 'pi': pi, 'logp': logp, 'mus': mus, 'stds': stds, 'pis': pis, 'logps': logps,
            'alpha': alpha, 'beta': beta, 'sample': sample}
    return out, meta


class Normalize:
    """per-file worker of `topaz normalize` / `topaz preprocess` (stats.py:277-331)"""

    def __init__(self, dest, scale, affine, num_iters, alpha, beta, sample, metadata, formats, use_cuda=True):
        self.dest, self.scale, self.affine, self.num_iters = dest, scale, affine, num_iters
        self.alpha, self.beta, self.sample, self.metadata, self.formats = alpha, beta, sample, metadata, formats

    def __call__(self, path):
        from .utils.image import downsample, load_image, save_image
        image = load_image(path, make_image=False)
        image, header, extended_header = image if type(image) is tuple else (image, None, None)
        x = image.astype(np.float32)
        if self.scale > 1:
            x = downsample(x, self.scale)
            if header:
                header = header._replace(ny=x.shape[0], nx=x.shape[1])
        x, metadata = normalize(x, alpha=self.alpha, beta=self.beta, num_iters=self.num_iters,
                                method='affine' if self.affine else 'gmm', sample=self.sample)
        name = os.path.splitext(os.path.basename(path))[0]
        base = os.path.join(self.dest, name)
        for f in self.formats:
            save_image(x, base, f=f, header=header, extended_header=extended_header)
        if self.metadata:
            md = dict(metadata)
            for k in ('mus', 'stds', 'pis', 'logps'):
                if k in md:
                    md[k] = np.asarray(md[k]).tolist()
            for k in ('mu', 'std', 'pi', 'logp'):
                if k in md:
                    md[k] = float(md[k])
            with open(base + '.metadata.json', 'w') as fh:
                json.dump(md, fh, indent=4)
        return name


def normalize_images(paths: List[str], dest: str, num_workers: int, scale: int, affine: bool, niters: int, alpha: float,
                     beta: float, sample: int, metadata: bool, formats: List[str], use_cuda: bool = True,
                     verbose: bool = False):
    """stats.py:334-352; the worker pool of the reference is not used: the fit runs on the GPU"""
    os.makedirs(dest, exist_ok=True)
    process = Normalize(dest, scale, affine, niters, alpha, beta, sample, metadata, formats)
    for path in paths:
        name = process(path)
        if verbose:
            print('# processed:', name, file=sys.stderr)
