"""Mirror of topaz/filters.py (gaussian_filter :6-19, inverse_filter :22-25, AffineFilter :28-37,
GaussianDenoise :51-80, InvGaussianFilter :83-96): 1->1 channel filters applied by the HIP
direct-convolution kernel (tpz_filter_2d)."""
from __future__ import annotations

import numpy as np
import torch

from . import runtime as rt


def gaussian_filter(sigma, s=11, dims=2):
    dim = s // 2
    r = np.arange(-dim, dim + 1)
    if dims == 2:
        xx, yy = np.meshgrid(r, r)
        d = xx ** 2 + yy ** 2
    else:
        xx, yy, zz = np.meshgrid(r, r, r)
        d = xx ** 2 + yy ** 2 + zz ** 2
    return np.exp(-0.5 * d / sigma ** 2)


def inverse_filter(w):
    F = np.fft.rfft2(np.fft.ifftshift(w))
    return np.fft.fftshift(np.fft.irfft2(1 / F, s=w.shape))


class _Filter2d:
    def __init__(self, weights: np.ndarray, use_cuda=True):
        self.weight = np.ascontiguousarray(weights, dtype=np.float32)
        self.bias = 0.0
        self.use_cuda = use_cuda

    def forward(self, x):
        return rt.filter_2d(x, self.weight, self.bias)

    __call__ = forward

    def apply(self, x):
        """numpy [H,W] -> numpy [H,W] (GaussianDenoise.apply, filters.py:71-80)"""
        if self.weight.ndim != 2:
            raise NotImplementedError('3-D Gaussian post-filter is not on the hot path')
        y = self.forward(torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)))
        return y.cpu().numpy()


class AffineFilter(_Filter2d):
    pass


class GaussianDenoise(_Filter2d):
    def __init__(self, sigma, scale=5, dims=2, use_cuda=True):
        width = 1 + 2 * int(np.ceil(sigma * scale))
        f = gaussian_filter(sigma, s=width, dims=dims)
        f /= f.sum()
        super().__init__(f, use_cuda)


class InvGaussianFilter(_Filter2d):
    def __init__(self, sigma, scale=5, use_cuda=True):
        width = 1 + 2 * int(np.ceil(sigma * scale))
        f = gaussian_filter(sigma, s=width)
        f /= f.sum()
        super().__init__(inverse_filter(f), use_cuda)
