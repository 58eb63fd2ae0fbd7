"""Mirror of topaz/algorithms.py non_maximum_suppression (:25-63) and non_maximum_suppression_3d
(:66-103) on the MI355X: same arguments, same (scores, coords) return, computed by the parallel
fix-point NMS of libtopaz_hip.so, bit-identical to the greedy loop (see csrc/nms.hip)."""
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
