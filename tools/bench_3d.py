#!/usr/bin/env python
"""Times tpz_denoise_3d on a few 192^3 tiles (C5 workload) -- profiling target for the 3-D kernels."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import synth_weights as sw  # noqa: E402
from topaz_amd.denoise import Denoise3D  # noqa: E402
from topaz_amd.denoising.models import DenoiseNet  # noqa: E402

d3 = Denoise3D(DenoiseNet('unet-3d', sw.unet_sd(13, nf=48, base_width=7, top_width=3, dims=3)))
t = torch.from_numpy(np.random.RandomState(2000).randn(96, 192, 192).astype(np.float32)).cuda()   # 1 x 2 x 2 tiles
d3.model.device_model.denoise_3d(t, 96, 48)
torch.cuda.synchronize()
t0 = time.perf_counter()
d3.model.device_model.denoise_3d(t, 96, 48)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f'4 tiles of 192^3: {dt * 1e3:.1f} ms -> {4 * 4.78 / dt:.1f} TFLOP/s')
