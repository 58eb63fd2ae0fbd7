#!/usr/bin/env python
"""NMS on the benchmark's own score map (resnet8-u64 logits of a 4096^2 N(0,1) micrograph, r = 14, t = -6): wall time
per call (device map in, pick table out) and the library's HIP-event time of its kernels."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import synth_weights as sw  # noqa: E402
from topaz_amd import runtime as rt  # noqa: E402


def main():
    ctx = rt.get_context(0)
    m = sw.hip_resnet('resnet8', 64, 7)[0]
    x = torch.from_numpy(np.random.RandomState(1000).randn(4096, 4096).astype(np.float32)).cuda()
    y = m(x[None, None])[0, 0]
    torch.cuda.synchronize()
    for r, thr in ((14, -6.0), (8, -6.0), (14, -3.0)):
        rt.nms(y, r, thr)
        torch.cuda.synchronize()
        ctx.prof_enable(True)
        ctx.prof_reset()
        t0 = time.perf_counter()
        for _ in range(5):
            s, c = rt.nms(y, r, thr)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        ms, n, _ = ctx.prof_get(3)
        ctx.prof_enable(False)
        print(f'r={r} t={thr}: {1e3 * dt:.2f} ms per call wall, {ms / 5:.2f} ms in the NMS bracket, {len(s)} picks, '
              f'{int((y > thr).sum())} candidates')


if __name__ == '__main__':
    main()
