"""C5 alone (BASELINE config 5: unet-3d on a 512x512x256 tomogram, 96/48 tiles), best of three runs -- the row of
tools/bench_configs.py that the A/B switches TPZ_NO_SRCMAJOR / TPZ_NO_POOL3D / TPZ_LANES (with TPZ_DEBUG=1) are measured on."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from tools import synth_weights as sw
from topaz_amd.denoise import Denoise3D
from topaz_amd.denoising.models import DenoiseNet
d3 = Denoise3D(DenoiseNet('unet-3d', sw.unet_sd(13, nf=48, base_width=7, top_width=3, dims=3)))
t = torch.from_numpy(np.random.RandomState(2000).randn(256, 512, 512).astype(np.float32)).cuda()
f = lambda: d3.model.device_model.denoise_3d(t, 96, 48)
f(); torch.cuda.synchronize()
ts = []
for _ in range(3):
    t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print('C5 denoise3d 512x512x256: %.1f ms (runs %s)' % (min(ts), ' '.join('%.1f' % x for x in ts)))
