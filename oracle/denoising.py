"""Oracle (test infrastructure): CPU restatement of the reference's denoising path.

Follows topaz/denoising/models.py (UDenoiseNet :74-175, UDenoiseNetSmall :178-244, DenoiseNet2
:52-66, UDenoiseNet3D :452-564, load_model :581-625), topaz/filters.py (AffineDenoise :40-48,
GaussianDenoise :51-80), topaz/denoise.py (Denoise._denoise :274-296, denoise_patches :299-324,
denoise :327-332, Denoise3D.denoise :340-377, denoise_image :382-416) and
topaz/denoising/datasets.py (PatchDataset :412-468).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

from .scoring import to_torch_sd


def _conv(x, sd, name, dims):
    w = sd[name + '.weight']
    f = F.conv3d if dims == 3 else F.conv2d
    return f(x, w, sd.get(name + '.bias'), padding=w.shape[-1] // 2)


def _pool(x, dims):
    return F.max_pool3d(x, 2) if dims == 3 else F.max_pool2d(x, 2)


def _up_cat(h, skip):
    # F.interpolate(h, size=skip.shape[2:], mode='nearest'); torch.cat([h, skip], 1)  (models.py:140-171)
    h = F.interpolate(h, size=tuple(skip.shape[2:]), mode='nearest')
    return torch.cat([h, skip], 1)


def unet_forward(x: torch.Tensor, sd: Dict[str, torch.Tensor], depth: int = 5, dims: int = 2, no_skip=(),
                 noise_only: bool = False) -> torch.Tensor:
    """UDenoiseNet (depth 5: enc1..enc6, dec5..dec1), UDenoiseNetSmall (depth 3: enc1..enc4, dec3..dec1)
    and UDenoiseNet3D (depth 5, dims 3).  x: [N,1,(D,)H,W].
    no_skip: decoder levels that upsample WITHOUT concatenating the skip tensor -- UDenoiseNet2 (models.py:321-338: dec2 and
    dec1).  noise_only: UDenoiseNet3 (models.py:447) returns x - dec1(h)."""
    lrelu = lambda t: F.leaky_relu(t, 0.1)
    skips = [x]
    h = x
    for i in range(1, depth + 1):
        h = _pool(lrelu(_conv(h, sd, f'enc{i}.0', dims)), dims)
        skips.append(h)
    h = lrelu(_conv(skips[-1], sd, f'enc{depth + 1}.0', dims))
    # decoder level L concatenates with skip p_{L-1}; dec1 concatenates with the input itself
    for lvl in range(depth, 0, -1):
        if lvl in no_skip:
            h = F.interpolate(h, size=tuple(skips[lvl - 1].shape[2:]), mode='nearest')
        else:
            h = _up_cat(h, skips[lvl - 1])
        h = lrelu(_conv(h, sd, f'dec{lvl}.0', dims))
        h = lrelu(_conv(h, sd, f'dec{lvl}.2', dims))
        if lvl == 1:
            h = _conv(h, sd, 'dec1.4', dims)       # no activation after the last conv
    return x - h if noise_only else h


def fcnn_forward(x, sd):
    # DenoiseNet2 (models.py:52-66): conv-lrelu-conv-lrelu-conv, all width x width, same padding
    h = F.leaky_relu(_conv(x, sd, 'net.0', 2), 0.1)
    h = F.leaky_relu(_conv(h, sd, 'net.2', 2), 0.1)
    return _conv(h, sd, 'net.4', 2)


def affine_forward(x, sd):
    # AffineDenoise / GaussianDenoise (filters.py:40-80): one 1->1 conv with same padding
    return _conv(x, sd, 'filter', 2)


def model_forward(kind: str, sd, x: torch.Tensor) -> torch.Tensor:
    if kind == 'unet':
        return unet_forward(x, sd, 5, 2)
    if kind == 'unet-small':
        return unet_forward(x, sd, 3, 2)
    if kind == 'unet2':
        return unet_forward(x, sd, 5, 2, no_skip=(2, 1))
    if kind == 'unet3':
        return unet_forward(x, sd, 5, 2, noise_only=True)
    if kind == 'unet-3d':
        return unet_forward(x, sd, 5, 3)
    if kind == 'fcnn':
        return fcnn_forward(x, sd)
    if kind == 'affine':
        return affine_forward(x, sd)
    raise ValueError(kind)


@torch.no_grad()
def denoise_whole(kind: str, sd, x: torch.Tensor) -> np.ndarray:
    """Denoise._denoise (denoise.py:274-296): torch mean / UNBIASED std of the array itself,
    normalise, model, un-normalise.  x: tensor of rank dims (2-D or 3-D)."""
    mu, std = x.mean(), x.std()
    xin = ((x - mu) / std)[None, None]
    pred = model_forward(kind, sd, xin).squeeze()
    pred = pred * std + mu
    return pred.numpy()


@torch.no_grad()
def denoise(kind: str, sd, x: np.ndarray, patch_size: int = -1, padding: int = 128) -> np.ndarray:
    """Denoise.denoise + denoise_patches (denoise.py:299-332) for 2-D images."""
    sd = to_torch_sd(sd)
    xt = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    s = patch_size + padding
    use_patch = (patch_size > 0) and (s < xt.shape[0] or s < xt.shape[1])
    if not use_patch:
        return denoise_whole(kind, sd, xt)
    y = np.zeros_like(x, dtype=np.float32)
    H, W = xt.shape
    for i in range(0, H, patch_size):
        for j in range(0, W, patch_size):
            si, ei = max(0, i - padding), min(H, i + patch_size + padding)
            sj, ej = max(0, j - padding), min(W, j + patch_size + padding)
            yij = denoise_whole(kind, sd, xt[si:ei, sj:ej])
            oi, oj = i - si, j - sj
            y[i:i + patch_size, j:j + patch_size] = yij[oi:oi + patch_size, oj:oj + patch_size]
    return y


@torch.no_grad()
def denoise3d(sd, tomo: np.ndarray, patch_size: int = 96, padding: int = 48) -> np.ndarray:
    """Denoise3D.denoise (denoise.py:340-377) with PatchDataset tiles (datasets.py:412-468):
    global numpy mean / POPULATION std; zero-filled (patch+2*pad)^3 tiles normalised globally,
    then per tile by _denoise (torch, unbiased); centre stitched."""
    sd = to_torch_sd(sd)
    tomo = np.ascontiguousarray(tomo, dtype=np.float32)
    denoised = np.zeros_like(tomo)
    mu, std = tomo.mean(), tomo.std()
    if patch_size < 1:
        denoised[:] = denoise_whole('unet-3d', sd, torch.from_numpy(tomo))
        return denoised
    D, H, W = tomo.shape
    d = patch_size + 2 * padding
    for i in range(0, D, patch_size):
        for j in range(0, H, patch_size):
            for k in range(0, W, patch_size):
                x = np.zeros((d, d, d), dtype=np.float32)
                si, ei = max(0, i - padding), min(D, i + patch_size + padding)
                sj, ej = max(0, j - padding), min(H, j + patch_size + padding)
                sk, ek = max(0, k - padding), min(W, k + patch_size + padding)
                sic, sjc, skc = padding - i + si, padding - j + sj, padding - k + sk
                x[sic:sic + ei - si, sjc:sjc + ej - sj, skc:skc + ek - sk] = tomo[si:ei, sj:ej, sk:ek]
                xt = (torch.from_numpy(x) - mu) / std
                # the DataLoader adds a batch dim of 1; _denoise's statistics run over the whole batch
                out = denoise_whole('unet-3d', sd, xt) * std + mu
                pz, py, px = denoised[i:i + patch_size, j:j + patch_size, k:k + patch_size].shape
                denoised[i:i + patch_size, j:j + patch_size, k:k + patch_size] = \
                    out[padding:padding + pz, padding:padding + py, padding:padding + px]
    return denoised


@torch.no_grad()
def denoise_image(kinds_sds, mic: np.ndarray, cutoff: float = 0, gaus_sigma: float = 0, patch_size: int = -1,
                  padding: int = 0, normalize: bool = False) -> np.ndarray:
    """denoise_image (denoise.py:382-416) without the lowpass / deconvolve branches (both crash
    in the reference): numpy mean / POPULATION std normalisation, optional pixel
    cutoff, optional Gaussian pre-filter, mean over the model ensemble, re-scale."""
    mu, std = mic.mean(), mic.std()
    x = (mic - mu) / std
    if cutoff > 0:
        x[(x < -cutoff) | (x > cutoff)] = 0
    if gaus_sigma > 0:
        f = gaussian_kernel(gaus_sigma)
        xt = torch.from_numpy(x.astype(np.float32))[None, None]
        x = F.conv2d(xt, torch.from_numpy(f)[None, None], torch.zeros(1), padding=f.shape[0] // 2).squeeze().numpy()
    out = sum(denoise(kind, sd, x, patch_size=patch_size, padding=padding) for kind, sd in kinds_sds) / len(kinds_sds)
    if normalize:
        out = (out - out.mean()) / out.std()
    else:
        out = std * out + mu
    return out


def gaussian_kernel(sigma: float, scale: float = 5, dims: int = 2) -> np.ndarray:
    """GaussianDenoise.__init__ (filters.py:55-59) + gaussian_filter (:6-19): width 1+2*ceil(sigma*scale),
    exp(-0.5 d^2/sigma^2), normalised to sum 1, cast to float32."""
    width = 1 + 2 * int(np.ceil(sigma * scale))
    dim = width // 2
    r = np.arange(-dim, dim + 1)
    if dims == 2:
        xx, yy = np.meshgrid(r, r)
        d = xx ** 2 + yy ** 2
    else:
        xx, yy, zz = np.meshgrid(r, r, r)
        d = xx ** 2 + yy ** 2 + zz ** 2
    f = np.exp(-0.5 * d / sigma ** 2)
    f /= f.sum()
    return f.astype(np.float32)


# ---------------------------------------------------------------------------------------------
# seeded synthetic weights (architectures whose pretrained blobs are absent): tools/synth_weights.py
# ---------------------------------------------------------------------------------------------
def synthetic_unet_sd(seed: int, nf: int = 48, base_width: int = 11, top_width: int = 5, depth: int = 5,
                      dims: int = 2) -> 'OrderedDict[str, np.ndarray]':
    from tools import synth_weights as sw
    return sw.unet_sd(seed, nf, base_width, top_width, depth, dims)


def downsample(x: np.ndarray, factor=1, shape=None) -> np.ndarray:
    """truncated-DFT downsample, restating topaz/utils/image.py:38-61: rfft2, keep the m x (n//2+1) block of
    low frequencies (rows 0..m//2-1 and the last ceil(m/2)), scale by (m*n)/(M*N), irfft2."""
    if shape is None:
        shape = (int(x.shape[-2] / factor), int(x.shape[-1] / factor))
    m, n = shape
    F = np.fft.rfft2(x)
    F = np.concatenate([F[..., 0:m // 2, 0:n // 2 + 1], F[..., -m // 2:, 0:n // 2 + 1]], axis=0)
    F = F * ((n * m) / (x.shape[-2] * x.shape[-1]))
    return np.fft.irfft2(F, s=shape).astype(x.dtype)
