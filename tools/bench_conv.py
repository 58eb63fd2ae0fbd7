#!/usr/bin/env python
"""Micro-benchmark of one fused convolution layer through libtopaz_hip.so (tuning aid).

    python tools/bench_conv.py --cin 128 --cout 256 --k 5 --dil 4 --size 2048 [--head] [--iters 5]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topaz_amd.runtime import DeviceModel, LayerProgram, get_context  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cin', type=int, default=128)
    ap.add_argument('--cout', type=int, default=256)
    ap.add_argument('--k', type=int, default=5)
    ap.add_argument('--dil', type=int, default=4)
    ap.add_argument('--pad', type=int, default=0)
    ap.add_argument('--size', type=int, default=2048)
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--head', action='store_true')
    a = ap.parse_args()
    ctx = get_context(0)
    rs = np.random.RandomState(0)
    P = LayerProgram(2)
    # a 1 -> cin stem (7x7 CIN1 kernel or direct) feeds the layer under test
    w0 = (rs.randn(a.cin, 1, 1, 1)).astype(np.float32)
    cur = P.conv(0, w0, None, slope=1.0) if a.cin > 1 else 0
    w = (rs.randn(a.cout, a.cin, a.k, a.k) / np.sqrt(a.cin * a.k * a.k)).astype(np.float32)
    kw = dict(head_w=rs.randn(a.cout).astype(np.float32), head_b=0.0) if a.head else {}
    P.conv(cur, w, rs.randn(a.cout).astype(np.float32), dil=a.dil, pad=a.pad, slope=0.0, **kw)
    m = DeviceModel(P, ctx)
    x = torch.randn(1, 1, a.size, a.size, device='cuda')
    m.forward(x)
    torch.cuda.synchronize()
    ctx.prof_enable(True)
    ctx.prof_reset()
    for _ in range(a.iters):
        m.forward(x)
    torch.cuda.synchronize()
    ms, n, fl = ctx.prof_get(0)
    # the stem (k=1, cin=1) is a direct/odd kernel; report the layer under test only
    span = a.dil * (a.k - 1)
    ho = a.size + 2 * a.pad - span
    flops = 2.0 * a.cout * a.cin * a.k * a.k * ho * ho
    print(f'conv cin={a.cin} cout={a.cout} k={a.k} dil={a.dil} out={ho}x{ho}: all mfma launches {ms / a.iters:.3f} ms/iter '
          f'({n // a.iters} launches), layer flops {flops / 1e12:.3f} TF -> >= {flops * a.iters / (ms * 1e-3) / 1e12:.1f} TFLOP/s '
          f'(lower bound: includes the stem launch if it ran on the MFMA kernel)')


if __name__ == '__main__':
    main()
