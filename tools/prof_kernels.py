#!/usr/bin/env python
"""Per-instantiation time and achieved TFLOP/s of the convolution kernels for one workload, patch lanes off, measured
with the library's own HIP-event profiler (tpz_prof_*):   python tools/prof_kernels.py [denoise|extract|denoise3d] [exact]
(exact: every convolution on the fp32 kernels, tpz_ctx_set_exact)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import synth_weights as sw  # noqa: E402
from topaz_amd import runtime as rt  # noqa: E402
from topaz_amd.denoise import Denoise, Denoise3D  # noqa: E402
from topaz_amd.denoising.models import DenoiseNet  # noqa: E402
from topaz_amd.model.classifier import LinearClassifier  # noqa: E402


def main(what, exact=False):
    ctx = rt.get_context(0)
    ctx.set_exact(exact)
    if what == 'denoise':
        x = torch.from_numpy(np.random.RandomState(1000).randn(4096, 4096).astype(np.float32)).cuda()
        dn = Denoise(DenoiseNet('unet', sw.unet_sd(11, nf=48, base_width=11, top_width=5)))
        fn = lambda: dn.denoise_device(x, 1024, 500)
    elif what in ('fcnn', 'unet-small', 'unet-v0.2.1'):
        x = torch.from_numpy(np.random.RandomState(1000).randn(4096, 4096).astype(np.float32)).cuda()
        dn = Denoise(what)
        fn = lambda: dn.denoise_device(x, 1024, 500)
    elif what in ('resnet8_u32', 'resnet16_u32'):
        from topaz_amd.model.factory import load_model
        x = torch.from_numpy(np.random.RandomState(1000).randn(4096, 4096).astype(np.float32)).cuda()
        m = load_model(what)
        m.eval(); m.fill(); m.cuda()
        fn = lambda: rt.nms(m(x[None, None])[0, 0], 14, -6.0)
    elif what == 'conv127':
        x = torch.from_numpy(np.random.RandomState(1000).randn(4096, 4096).astype(np.float32)).cuda()
        m = LinearClassifier('conv127', sw.basic_sd((7, 5, 5, 5, 5), 32, 7))
        m.eval(); m.fill(); m.cuda()
        fn = lambda: rt.nms(m(x[None, None])[0, 0], 14, -6.0)
    elif what == 'extract':
        x = torch.from_numpy(np.random.RandomState(1000).randn(4096, 4096).astype(np.float32)).cuda()
        m = sw.hip_resnet('resnet8', 64, 7)[0]
        m.eval(); m.fill(); m.cuda()
        fn = lambda: rt.nms(m(x[None, None])[0, 0], 14, -6.0)
    else:
        d3 = Denoise3D(DenoiseNet('unet-3d', sw.unet_sd(13, nf=48, base_width=7, top_width=3, dims=3)))
        t = torch.from_numpy(np.random.RandomState(2000).randn(192, 192, 384).astype(np.float32)).cuda()
        fn = lambda: d3.model.device_model.denoise_3d(t, 96, 48)
    fn()
    torch.cuda.synchronize()
    ctx.set_lanes(False)            # kernels timed in isolation: no concurrent patch streams
    ctx.prof_enable(True)
    ctx.prof_reset()
    fn()
    torch.cuda.synchronize()
    rows = ctx.prof_kernels()
    ctx.prof_enable(False)
    tot = sum(r[1] for r in rows)
    print(f'# {what}{" (exact fp32)" if exact else ""}: {tot:.2f} ms in conv_mfma kernels, {sum(r[3] for r in rows) / 1e12:.2f} TFLOP executed')
    print(f'{"calls":>6} {"total_ms":>10} {"avg_ms":>9} {"TFLOP/s":>8} {"pct":>6}  kernel')
    for name, ms, n, fl in rows:
        print(f'{n:6d} {ms:10.3f} {ms / n:9.4f} {fl / ms / 1e9:8.1f} {100 * ms / tot:6.2f}  {name}')
    for cls, label in ((1, 'conv_direct'), (2, 'elementwise'), (3, 'nms')):
        ms, n, _ = ctx.prof_get(cls)
        print(f'{n:6d} {ms:10.3f} {"":>9} {"":>8} {"":>6}  [{label}]')


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else 'denoise', 'exact' in sys.argv[2:])
