#!/usr/bin/env python
"""Two scoring passes of the bench's ResNet8-u64 over a 4096^2 micrograph and nothing else: the light command the
`rocprofv3 --pmc` passes (FETCH_SIZE, WRITE_SIZE, TCC_HIT / TCC_MISS: one counter set per pass) are run on."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import synth_weights as sw  # noqa: E402

x = torch.from_numpy(np.random.RandomState(1000).randn(4096, 4096).astype(np.float32)).cuda()
m = sw.hip_resnet('resnet8', 64, 7)[0]
m.eval(); m.fill(); m.cuda()
for _ in range(2):
    y = m(x[None, None])
torch.cuda.synchronize()
print('ok', float(y.abs().max()))
