"""Mirror of the patching helpers of topaz/model/utils.py: insize_from_outsize (:39-68),
predict_in_patches (:110-130), get_patches (:133-169), reconstruct_from_patches (:172-193).

Patch geometry is host logic; predict_in_patches keeps the image, the tiles and the stitched map on the device
(one upload, one download).  Quirk kept from the reference: all-zero tiles are skipped by get_patches (:159,165)
while reconstruct_from_patches (:181-191) does not know about it, so an image with an all-zero tile ends in
IndexError exactly as upstream does.
"""
from __future__ import annotations

from typing import List

import numpy as np
import torch


def insize_from_outsize(layers, outsize):
    """layers: objects (or dicts) with kernel_size / stride / padding / dilation"""
    for layer in layers[::-1]:
        get = (lambda k, d: layer.get(k, d)) if isinstance(layer, dict) else (lambda k, d: getattr(layer, k, d))
        first = lambda v: v[0] if isinstance(v, tuple) else v
        k, s = first(get('kernel_size', 1)), first(get('stride', 1))
        p, d = first(get('padding', 0)), first(get('dilation', 1))
        outsize = (outsize - 1) * s + 1 + (k - 1) * d - 2 * p
    return outsize


def get_patches(X: torch.Tensor, patch_size: int, patch_padding: int = 0, is_3d: bool = False) -> List[torch.Tensor]:
    y, x = X.shape[-2:]
    z = X.shape[-3] if is_3d else None
    pad = (patch_padding, patch_padding) * (3 if is_3d else 2)
    X = torch.nn.functional.pad(X, pad)
    y_pad, x_pad = X.shape[-2:]
    z_pad = X.shape[-3] if is_3d else None
    step = patch_size - 2 * patch_padding
    patches = []
    for i in range(0, y, step):
        for j in range(0, x, step):
            i_end, j_end = min(i + patch_size, y_pad), min(j + patch_size, x_pad)
            if is_3d:
                for k in range(0, z, step):
                    k_end = min(k + patch_size, z_pad)
                    patch = X[..., k:k_end, i:i_end, j:j_end]
                    if patch.abs().sum() == 0:
                        continue
                    patches.append(patch)
            else:
                patch = X[..., i:i_end, j:j_end]
                if patch.abs().sum() == 0:
                    continue
                patches.append(patch)
    return patches


def reconstruct_from_patches(patches, original_shape, patch_size, patch_padding=0, is_3d=False) -> np.ndarray:
    y, x = original_shape[-2:]
    z = original_shape[-3] if is_3d else None
    step = patch_size - patch_padding * 2
    out = np.zeros(original_shape)                 # float64, like the reference
    idx = 0
    for i in range(0, y, step):
        for j in range(0, x, step):
            if is_3d:
                for k in range(0, z, step):
                    p = patches[idx]
                    out[..., k:k + p.shape[-3], i:i + p.shape[-2], j:j + p.shape[-1]] = p
                    idx += 1
            else:
                p = patches[idx]
                out[..., i:i + p.shape[-2], j:j + p.shape[-1]] = p
                idx += 1
    return out


def predict_in_patches(model, X: torch.Tensor, patch_size: int, is_3d: bool = False, use_cuda: bool = True) -> np.ndarray:
    """X: [1,1,(D,)H,W] host or device tensor -> float64 array of the same shape (model/utils.py:110-130): tiles of
    `patch_size` cut from the image padded by width // 2 at stride patch_size - 2 * (width // 2); the filled model pads each
    tile again, its scores are cropped by width // 2 and stitched.
    The image goes to the device ONCE; tiles are device views, the stitched map is assembled on the device and comes back in
    one copy (upstream moves every tile on and off the GPU).  The all-zero-tile quirk is kept: get_patches skips such tiles
    (:159,165) while reconstruct_from_patches does not know about it and runs out of tiles -- the same IndexError here."""
    pad = model.width // 2
    dims = 3 if is_3d else 2
    dev = model.device_model.ctx.torch_device()
    x = X.to(device=dev, dtype=torch.float32)
    padded = torch.nn.functional.pad(x, (pad, pad) * dims)
    shape = tuple(x.shape[-dims:])
    step = patch_size - 2 * pad
    if step <= 0:
        raise ValueError(f'patch_size {patch_size} does not exceed the receptive field {model.width}')
    starts = [(i, j, k) for i in range(0, shape[-2], step) for j in range(0, shape[-1], step)
              for k in (range(0, shape[-3], step) if is_3d else (None,))]
    scored = []
    for (i, j, k) in starts:
        tile = padded[..., i:i + patch_size, j:j + patch_size] if k is None else \
            padded[..., k:k + patch_size, i:i + patch_size, j:j + patch_size]
        if float(tile.abs().sum()) == 0:
            continue                                       # get_patches drops all-zero tiles
        with torch.no_grad():
            s = model(tile.contiguous())[0, 0]
        s = s[..., pad:-pad, pad:-pad]
        scored.append(s[pad:-pad] if is_3d else s)
    if len(scored) < len(starts):
        # upstream stitches patches[idx] for every slot of the grid and runs past the end of the kept tiles
        raise IndexError('list index out of range (an all-zero tile was skipped: topaz/model/utils.py:159,181-191)')
    out = torch.zeros(tuple(X.shape), dtype=torch.float64, device=dev)
    for (i, j, k), s in zip(starts, scored):
        if k is None:
            out[..., i:i + s.shape[-2], j:j + s.shape[-1]] = s
        else:
            out[..., k:k + s.shape[-3], i:i + s.shape[-2], j:j + s.shape[-1]] = s
    return out.cpu().numpy()
