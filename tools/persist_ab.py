#!/usr/bin/env python
"""Same-process A/B of tpz_ctx_set_persist on the scoring stage of the bench (ResNet8-u64 at 4096^2 + NMS): mode 1 (the default:
persistent workgroups on the tiles of up to 96 channels) against mode 2 (wherever the kernel supports them: also the 128-channel
8-wave 3x3 tiles) and mode 0 (never), alternating, with the library's own per-kernel HIP-event times."""
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
    x = torch.from_numpy(np.random.RandomState(1000).randn(4096, 4096).astype(np.float32)).cuda()
    m = sw.hip_resnet('resnet8', 64, seed=7)[0]
    fn = lambda: rt.nms(m(x[None, None])[0, 0], 14, -6.0)
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ref = None
    for rep in range(3):
        for mode in (1, 2, 0):
            ctx.set_persist(mode)
            fn(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                out = fn()
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / 4
            if ref is None:
                ref = out
            same = torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])
            ctx.prof_enable(1); ctx.prof_reset(); fn(); torch.cuda.synchronize()
            rows = sorted(ctx.prof_kernels(), key=lambda r: -r[1])[:6]
            ctx.prof_enable(False)
            print(f'rep {rep} persist mode {mode}: {ms:7.2f} ms per micrograph   identical picks: {same}')
            for nm, kms, n, fl in rows:
                print(f'      {kms:7.3f} ms  {fl / kms / 1e9:6.1f} TF/s  {nm}')
    ctx.set_persist(1)


if __name__ == '__main__':
    main()
