"""Mirror of topaz/algorithms.py non_maximum_suppression (:25-63) and non_maximum_suppression_3d
(:66-103) on the MI355X: same arguments, same (scores, coords) return, computed by the parallel
fix-point NMS of libtopaz_hip.so, bit-identical to the greedy loop (see csrc/nms.hip); plus match_coordinates (:7-22),
the pick-to-target assignment of the `--targets` validation (host code: a few hundred points per micrograph)."""
from __future__ import annotations

from typing import Tuple

import numpy as np
import torch

from . import runtime as rt


def non_maximum_suppression(x, r: int, threshold: float = -np.inf) -> Tuple[np.ndarray, np.ndarray]:
    """x: [H,W] score map (numpy array or tensor, host or device).  Returns numpy scores[n] (fp32,
    descending) and coords[n,2] int32 as (x, y)."""
    s, c = rt.nms(x if torch.is_tensor(x) else np.asarray(x), int(r), float(threshold))
    return s.cpu().numpy(), c.cpu().numpy()


def non_maximum_suppression_3d(x, r: int, scale: float = 1.0, threshold: float = -np.inf):
    """x: [D,H,W]; coords[n,3] as (x, y, z)."""
    s, c = rt.nms(x if torch.is_tensor(x) else np.asarray(x), int(r), float(threshold), scale=float(scale))
    return s.cpu().numpy(), c.cpu().numpy()


def match_coordinates(targets: np.ndarray, preds: np.ndarray, radius: float) -> Tuple[np.ndarray, np.ndarray]:
    """One-to-one assignment of predicted to labelled coordinates that maximises the total gain r^2 - d^2 over pairs closer
    than `radius` (pairs further apart gain nothing): the linear assignment problem of topaz/algorithms.py:7-22, solved on
    the same full pred x target cost matrix so that ties resolve like upstream.  Returns (matched[n_pred] in {0, 1} fp32,
    dist[n_pred] = distance of each prediction to the target the solver paired it with, 0 where it got none)."""
    from scipy.optimize import linear_sum_assignment
    from scipy.spatial.distance import cdist
    preds = np.asarray(preds, dtype=np.float64).reshape(len(preds), -1)
    targets = np.asarray(targets, dtype=np.float64).reshape(len(targets), preds.shape[1] if len(preds) else -1)
    matched = np.zeros(len(preds), dtype=np.float32)
    dist = np.zeros(len(preds))
    if len(preds) == 0 or len(targets) == 0:
        return matched, dist
    d2 = cdist(preds, targets, 'sqeuclidean')
    gain = np.minimum(d2 - float(radius) ** 2, 0.0)
    rows, cols = linear_sum_assignment(gain)
    dist[rows] = np.sqrt(d2[rows, cols])
    matched[rows[gain[rows, cols] < 0]] = 1
    return matched, dist
