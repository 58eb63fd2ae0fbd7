"""Mirror of topaz/predict.py:7-35 (batches / score_stream / score)."""
from typing import Iterable, Iterator, List

import numpy as np
import torch


def batches(X: Iterable[np.ndarray], batch_size: int = 1) -> Iterator[torch.Tensor]:
    batch = []
    for x in X:
        batch.append(torch.from_numpy(np.ascontiguousarray(x)).float())
        if len(batch) >= batch_size:
            yield torch.stack(batch, 0)
            batch = []
    if len(batch) > 0:
        yield torch.stack(batch, 0)


def score_stream(model, images: Iterable[np.ndarray], use_cuda: bool = True, batch_size: int = 1) -> Iterator[np.ndarray]:
    with torch.no_grad():
        for x in batches(images, batch_size=batch_size):
            logits = model(x.unsqueeze(1).cuda()).squeeze(1).cpu().numpy()
            for i in range(len(logits)):
                yield logits[i]


def score(model, images: Iterable[np.ndarray], use_cuda: bool = True, batch_size: int = 1) -> List[np.ndarray]:
    return list(score_stream(model, images, use_cuda=use_cuda, batch_size=batch_size))
