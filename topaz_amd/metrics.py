"""average_precision, restating topaz/metrics.py:53-89 (bucketed AP over tied predictions); used only by
the `--targets` radius search of `topaz extract`."""
import numpy as np


def average_precision(target, pred, N=None):
    n = target.sum() if N is None else N
    order = np.argsort(-np.asarray(pred, dtype=np.float32), kind='stable')
    p = -np.asarray(pred, dtype=np.float32)[order]
    t = np.asarray(target, dtype=np.float32)[order]
    mask = np.zeros(len(p), dtype=bool)
    mask[:-1] = p[:-1] != p[1:]
    mask[-1] = True
    last = np.where(mask)[0] + 1                      # predicted positives at each bucket
    tp_cum = np.cumsum(t)[mask]
    r = np.diff(np.concatenate([[0], tp_cum.astype(int)]))
    pr = tp_cum.astype(int) / last
    return np.sum(pr * r) / n
