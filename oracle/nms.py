"""Oracle (test infrastructure): greedy non-maximum suppression, restated from
topaz/algorithms.py:25-63 (2-D) and :66-103 (3-D).

`nms2d_py` / `nms3d_py` are line-by-line numpy/python restatements (small inputs only);
`nms2d` / `nms3d` call the C restatement (oracle/nms_c.c) and are what the parity tests and the
CPU baseline use at full size.  Tie order among equal scores is fixed to "stable argsort,
reversed" = descending flat index (the reference's `np.argsort(A)[::-1]` leaves it undefined).
"""
from __future__ import annotations

import ctypes as C
from typing import Tuple

import numpy as np

from . import build as _build

_lib = None


def _load():
    global _lib
    if _lib is None:
        lib = C.CDLL(_build.build())
        lib.nms2d_oracle.restype = C.c_long
        lib.nms2d_oracle.argtypes = [C.c_void_p, C.c_long, C.c_long, C.c_long, C.c_float, C.c_void_p, C.c_void_p]
        lib.nms3d_oracle.restype = C.c_long
        lib.nms3d_oracle.argtypes = [C.c_void_p, C.c_long, C.c_long, C.c_long, C.c_double, C.c_float, C.c_void_p,
                                     C.c_void_p]
        _lib = lib
    return _lib


def _thr(t) -> float:
    return float(np.float32(t)) if np.isfinite(t) else float(t)


def nms2d(x: np.ndarray, r: int, threshold: float = -np.inf) -> Tuple[np.ndarray, np.ndarray]:
    x = np.ascontiguousarray(x, dtype=np.float32)
    H, W = x.shape
    scores = np.zeros(x.size, dtype=np.float32)
    coords = np.zeros((x.size, 2), dtype=np.int32)
    n = _load().nms2d_oracle(x.ctypes.data, H, W, int(r), _thr(threshold), scores.ctypes.data, coords.ctypes.data)
    return scores[:n].copy(), coords[:n].copy()


def nms3d(x: np.ndarray, r: int, scale: float = 1.0, threshold: float = -np.inf) -> Tuple[np.ndarray, np.ndarray]:
    x = np.ascontiguousarray(x, dtype=np.float32)
    D, H, W = x.shape
    scores = np.zeros(x.size, dtype=np.float32)
    coords = np.zeros((x.size, 3), dtype=np.int32)
    n = _load().nms3d_oracle(x.ctypes.data, D, H, W, float(scale * r), _thr(threshold), scores.ctypes.data,
                             coords.ctypes.data)
    return scores[:n].copy(), coords[:n].copy()


def _order(A: np.ndarray, ties: str) -> np.ndarray:
    if ties == 'desc_index':          # stable argsort, reversed
        return np.argsort(A, axis=None, kind='stable')[::-1]
    if ties == 'asc_index':           # the other extreme, used to test tie-insensitivity of a fixture
        return np.argsort(-A, axis=None, kind='stable')
    raise ValueError(ties)


def nms2d_py(x: np.ndarray, r: int, threshold: float = -np.inf, ties: str = 'desc_index'):
    """algorithms.py:25-63 with an explicit tie order"""
    width = r
    ii, jj = np.meshgrid(np.arange(-width, width + 1), np.arange(-width, width + 1))
    mask = (ii ** 2 + jj ** 2) <= r * r
    ii, jj = ii[mask], jj[mask]
    major_axis = x.shape[1]
    A = x.ravel()
    I = _order(A, ties)
    S = set()
    scores, coords = [], []
    for i in I:
        if A[i] <= threshold:
            break
        if i not in S:
            xx, yy = i % major_axis, i // major_axis
            scores.append(A[i])
            coords.append((xx, yy))
            y_coords = np.clip(yy + ii, 0, x.shape[0])
            x_coords = np.clip(xx + jj, 0, x.shape[1])
            S.update((y_coords * major_axis + x_coords).tolist())
    return np.asarray(scores, dtype=np.float32), np.asarray(coords, dtype=np.int32).reshape(-1, 2)


def nms3d_py(x: np.ndarray, r: int, scale: float = 1.0, threshold: float = -np.inf, ties: str = 'desc_index'):
    """algorithms.py:66-103 with an explicit tie order"""
    r = scale * r
    width = int(np.ceil(r))
    a = np.arange(-width, width + 1)
    ii, jj, kk = np.meshgrid(a, a, a)
    mask = (ii ** 2 + jj ** 2 + kk ** 2) <= r * r
    ii, jj, kk = ii[mask], jj[mask], kk[mask]
    zstride, ystride = x.shape[1] * x.shape[2], x.shape[2]
    deltas = (ii * zstride + jj * ystride + kk).tolist()
    A = x.ravel()
    I = _order(A, ties)
    S = set()
    scores, coords = [], []
    for i in I:
        if A[i] <= threshold:
            break
        if i not in S:
            zz, yy, xx = np.unravel_index(i, x.shape)
            scores.append(A[i])
            coords.append((xx, yy, zz))
            S.update(int(i) + d for d in deltas)
    return np.asarray(scores, dtype=np.float32), np.asarray(coords, dtype=np.int32).reshape(-1, 3)
