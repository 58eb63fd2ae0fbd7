#!/usr/bin/env python
"""Times the individual BASELINE.json configs on one MI355X (numbers quoted in DESIGN.md section 5):
C2 ResNet8 extract 4096^2 (u64 seeded, u32 pretrained), C3 U-Net 2-D denoise 4096^2 (default patching and
whole image), C5 unet-3d on a 512x512x256 tomogram, plus ResNet16-u64 (the `topaz extract` default) and conv127."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import synth_weights as sw  # noqa: E402
from topaz_amd import runtime as rt  # noqa: E402
from topaz_amd.denoise import Denoise, Denoise3D  # noqa: E402
from topaz_amd.denoising.models import DenoiseNet  # noqa: E402
from topaz_amd.model.classifier import LinearClassifier  # noqa: E402
from topaz_amd.model.factory import load_model  # noqa: E402


def timed(fn, n=2):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    ctx = rt.get_context(0)
    x = torch.from_numpy(np.random.RandomState(1000).randn(4096, 4096).astype(np.float32)).cuda()
    rows = []

    def scorer(m):
        m.eval(); m.fill(); m.cuda()
        return lambda: rt.nms(m(x[None, None])[0, 0], 14, -6.0)

    rows.append(('C2 extract resnet8 u64 (seeded) + NMS', timed(scorer(sw.hip_resnet('resnet8', 64, 7)[0])), 44.23))
    rows.append(('C2 extract resnet8_u32 (pretrained) + NMS', timed(scorer(load_model('resnet8_u32'))), 11.09))
    rows.append(('extract resnet16 u64 (seeded, CLI default) + NMS', timed(scorer(sw.hip_resnet('resnet16', 64, 7)[0])), 61.99))
    rows.append(('extract conv127 u32 (seeded) + NMS', timed(scorer(LinearClassifier('conv127', sw.basic_sd((7, 5, 5, 5, 5), 32, 7)))), 3.61))
    dn = Denoise(DenoiseNet('unet', sw.unet_sd(11, nf=48, base_width=11, top_width=5)))
    rows.append(('C3 denoise unet b11/t5 (seeded), -s 1024 -p 500', timed(lambda: dn.denoise_device(x, 1024, 500)), 29.06))
    rows.append(('C3 denoise unet b11/t5 (seeded), whole image', timed(lambda: dn.denoise_device(x, -1, 0)), 9.68))
    d21 = Denoise('unet-v0.2.1')
    rows.append(('C3 denoise unet-v0.2.1 (pretrained), -s 1024 -p 500', timed(lambda: d21.denoise_device(x, 1024, 500)), None))
    d3 = Denoise3D(DenoiseNet('unet-3d', sw.unet_sd(13, nf=48, base_width=7, top_width=3, dims=3)))
    t = torch.from_numpy(np.random.RandomState(2000).randn(256, 512, 512).astype(np.float32)).cuda()
    rows.append(('C5 denoise3d unet-3d nf48 (seeded) 512x512x256, 96/48 tiles', timed(lambda: d3.model.device_model.denoise_3d(t, 96, 48), 1), 516.7))
    # 3-D scoring (`extract --dims 3`): a filled 3-D ResNet8 of 32 units over a 128^3 volume, on the plane-stacked 2xf16 kernels
    # (default) and on the fp32 kernels
    sd3 = sw.calibrate_head(sw.resnet_sd_uncalibrated('resnet8', 32, 7, dims=3), (1.0, 0.0))
    m3 = LinearClassifier('resnet8', sd3)
    m3.eval(); m3.fill(); m3.cuda()
    v = torch.from_numpy(np.random.RandomState(3000).randn(128, 128, 128).astype(np.float32)).cuda()
    tf3 = sum(2.0 * w.size * v.numel() for k, w in sd3.items() if k.endswith('weight') and w.ndim == 5) / 1e12
    rows.append(('extract --dims 3 resnet8 u32 (seeded) 128^3, 2xf16', timed(lambda: m3(v[None, None])), tf3))
    ctx.set_exact(True)
    rows.append(('extract --dims 3 resnet8 u32 (seeded) 128^3, fp32 kernels', timed(lambda: m3(v[None, None])), tf3))
    ctx.set_exact(False)
    print(f'{"config":<62} {"ms":>10} {"alg. TFLOP":>11} {"TFLOP/s":>9}')
    for name, ms, tf in rows:
        print(f'{name:<62} {ms:10.1f} {tf if tf else float("nan"):11.2f} {(tf / ms * 1e3) if tf else float("nan"):9.1f}')


if __name__ == '__main__':
    main()
