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
            return self._apply_volume(x)
        y = self.forward(torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)))
        return y.cpu().numpy()

    def _apply_volume(self, x):
        raise NotImplementedError('only the (separable) Gaussian has a 3-D form')


class AffineFilter(_Filter2d):
    pass


class GaussianDenoise(_Filter2d):
    def __init__(self, sigma, scale=5, dims=2, use_cuda=True):
        width = 1 + 2 * int(np.ceil(sigma * scale))
        f = gaussian_filter(sigma, s=width, dims=dims)
        f /= f.sum()
        super().__init__(f, use_cuda)
        g = gaussian_filter(sigma, s=width, dims=2)[width // 2]        # the 1-D factor: exp(-r^2 / 2 sigma^2)
        self._line = (g / g.sum()).astype(np.float64)

    def _apply_volume(self, x):
        """numpy [D,H,W] -> numpy [D,H,W]: the Conv3d(1, 1, width, padding=width//2) of filters.py:63-64.  The normalised
        3-D Gaussian is the outer product of three normalised 1-D Gaussians, so the volume takes three zero-padded line
        passes on the device -- along x on [D*H, W], along y on [H, D*W], along z on [D, H*W] -- each one a
        tpz_filter_2d call whose square kernel has a single non-zero row or column."""
        ctx = rt.get_context()
        v = rt.as_device_f32(torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)), ctx)
        D, H, W = v.shape
        k = self._line.shape[0]
        row = np.zeros((k, k), np.float32)
        row[k // 2, :] = self._line
        col = np.ascontiguousarray(row.T)
        v = rt.filter_2d(v.reshape(D * H, W), row, 0.0, ctx).reshape(D, H, W)
        v = rt.filter_2d(v.permute(1, 0, 2).contiguous().reshape(H, D * W), col, 0.0, ctx).reshape(H, D, W).permute(1, 0, 2)
        v = rt.filter_2d(v.contiguous().reshape(D, H * W), col, 0.0, ctx).reshape(D, H, W)
        return v.cpu().numpy()


class InvGaussianFilter(_Filter2d):
    def __init__(self, sigma, scale=5, use_cuda=True):
        width = 1 + 2 * int(np.ceil(sigma * scale))
        f = gaussian_filter(sigma, s=width)
        f /= f.sum()
        super().__init__(inverse_filter(f), use_cuda)
