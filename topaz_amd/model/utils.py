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


def tile_origins(shape, step: int):
    """origins of the tile grid over an image / volume of `shape` = ([D,] H, W): rows outermost, then columns, then -- for a
    volume -- planes, which is the order both the cutting and the stitching side walk (model/utils.py:151-166, 181-191)"""
    if len(shape) == 2:
        return [(None, i, j) for i in range(0, shape[0], step) for j in range(0, shape[1], step)]
    return [(k, i, j) for i in range(0, shape[1], step) for j in range(0, shape[2], step) for k in range(0, shape[0], step)]


def _window(t, origin, size):
    """the (at most) size^dims window of tensor / array `t` at `origin`, clipped by slicing at the far edges"""
    k, i, j = origin
    if k is None:
        return t[..., i:i + size, j:j + size]
    return t[..., k:k + size, i:i + size, j:j + size]


def get_patches(X: torch.Tensor, patch_size: int, patch_padding: int = 0, is_3d: bool = False) -> List[torch.Tensor]:
    """tiles of `patch_size` cut from X zero-padded by `patch_padding` on every side, at stride patch_size - 2 * padding
    over the UNPADDED extent (model/utils.py:133-169).  Tiles that hold nothing but zeros are left out, as upstream."""
    dims = 3 if is_3d else 2
    padded = torch.nn.functional.pad(X, (patch_padding,) * (2 * dims))
    tiles = (_window(padded, o, patch_size) for o in tile_origins(tuple(X.shape[-dims:]), patch_size - 2 * patch_padding))
    return [t for t in tiles if bool(t.ne(0).any())]


def reconstruct_from_patches(patches, original_shape, patch_size, patch_padding=0, is_3d=False) -> np.ndarray:
    """float64 array of `original_shape` with patch n pasted at the n-th origin of the same grid (model/utils.py:172-193);
    a patch list shortened by get_patches' zero-tile rule runs out here (IndexError), as upstream."""
    dims = 3 if is_3d else 2
    canvas = np.zeros(original_shape)
    for n, origin in enumerate(tile_origins(tuple(original_shape[-dims:]), patch_size - 2 * patch_padding)):
        piece = patches[n]
        k, i, j = origin
        if k is None:
            canvas[..., i:i + piece.shape[-2], j:j + piece.shape[-1]] = piece
        else:
            canvas[..., k:k + piece.shape[-3], i:i + piece.shape[-2], j:j + piece.shape[-1]] = piece
    return canvas


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
    starts = tile_origins(shape, step)
    scored = []
    for origin in starts:
        tile = _window(padded, origin, patch_size)
        if not bool(tile.ne(0).any()):
            continue                                       # get_patches drops all-zero tiles
        with torch.no_grad():
            s = model(tile.contiguous())[0, 0]
        s = s[..., pad:-pad, pad:-pad]
        scored.append(s[pad:-pad] if is_3d else s)
    if len(scored) < len(starts):
        # upstream stitches patches[idx] for every slot of the grid and runs past the end of the kept tiles
        raise IndexError('list index out of range (an all-zero tile was skipped: topaz/model/utils.py:159,181-191)')
    out = torch.zeros(tuple(X.shape), dtype=torch.float64, device=dev)
    for (k, i, j), s in zip(starts, scored):
        if k is None:
            out[..., i:i + s.shape[-2], j:j + s.shape[-1]] = s
        else:
            out[..., k:k + s.shape[-3], i:i + s.shape[-2], j:j + s.shape[-1]] = s
    return out.cpu().numpy()
