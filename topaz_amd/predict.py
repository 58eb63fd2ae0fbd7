"""Batched scoring helpers with the call surface of topaz/predict.py:7-35 (`batches`, `score_stream`,
`score`).  The HIP model scores the images of a batch one after another on the device, so batching only
groups the host-to-device copies."""
from __future__ import annotations

from itertools import islice
from typing import Iterable, Iterator, List

import numpy as np
import torch


def batches(X: Iterable[np.ndarray], batch_size: int = 1) -> Iterator[torch.Tensor]:
    """stack consecutive images into fp32 tensors [<=batch_size, H, W]"""
    it = iter(X)
    while True:
        group = [torch.as_tensor(np.array(x, dtype=np.float32)) for x in islice(it, max(1, batch_size))]
        if not group:
            return
        yield torch.stack(group)


def score_stream(model, images: Iterable[np.ndarray], use_cuda: bool = True, batch_size: int = 1) -> Iterator[np.ndarray]:
    """yield one [H, W] logit map per image"""
    for group in batches(images, batch_size):
        with torch.no_grad():
            logits = model(group[:, None].cuda())[:, 0]
        yield from logits.cpu().numpy()


def score(model, images: Iterable[np.ndarray], use_cuda: bool = True, batch_size: int = 1) -> List[np.ndarray]:
    return list(score_stream(model, images, use_cuda=use_cuda, batch_size=batch_size))
