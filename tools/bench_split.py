#!/usr/bin/env python
"""fp32-MFMA vs 2xf16 kernels, layer by layer, on the ResNet8-u64 scoring stack over a 4096^2 micrograph
(per-instantiation times from the library's HIP-event profiler)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import synth_weights as sw  # noqa: E402
from topaz_amd import runtime as rt  # noqa: E402
from topaz_amd.model.classifier import LinearClassifier  # noqa: E402


def main(arch='resnet8', size=4096):
    ctx = rt.get_context(0)
    x = torch.from_numpy(np.random.RandomState(1000).randn(size, size).astype(np.float32)).cuda()
    m = sw.hip_resnet(arch, 64, 7)[0]
    m.eval(); m.fill(); m.cuda()
    outs = {}
    for exact in (True, False):
        ctx.set_exact(exact)
        y = m(x[None, None])
        torch.cuda.synchronize()
        ctx.prof_enable(True)
        ctx.prof_reset()
        y = m(x[None, None])
        torch.cuda.synchronize()
        rows = ctx.prof_kernels()
        ctx.prof_enable(False)
        outs[exact] = y
        tot = sum(r[1] for r in rows)
        print(f'# {arch} {"fp32 MFMA" if exact else "2xf16"}: {tot:.2f} ms in conv kernels, '
              f'{sum(r[3] for r in rows) / 1e12:.2f} TFLOP (fp32-equivalent) -> {sum(r[3] for r in rows) / tot / 1e9:.1f} TFLOP/s')
        for name, ms, n, fl in rows:
            print(f'{n:6d} {ms:10.3f} {ms / n:9.4f} {fl / ms / 1e9:8.1f} {100 * ms / tot:6.2f}  {name}')
    d = (outs[True] - outs[False]).abs() / (1 + outs[True].abs())
    print(f'max scaled difference fp32 vs 2xf16: {d.max().item():.3e}; split stats {m.device_model.split_stats()}')


if __name__ == '__main__':
    main(*(sys.argv[1:2]), *(int(a) for a in sys.argv[2:3]))
