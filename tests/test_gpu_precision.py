"""How close is the default (2xf16) path to the EXACT result, measured instead of claimed.

The 2xf16 kernels multiply 22-bit operands (hi + lo f16 halves, three exact products per multiply-add, fp32 accumulation);
torch's fp32 convolution multiplies 24-bit operands.  DESIGN.md section 3.1 says the end-to-end error of the split path is that of
an fp32 evaluation (summation order, not operand width).  These tests evaluate the benchmark's own networks in float64
(oracle with dtype=torch.float64: same fp32 weights, same fp32 input, double arithmetic) and bound

    |2xf16 - f64|  <=  RATIO * |torch fp32 - f64|          (RATIO = 2, the bar VERDICT round 2 asked for)

next to the absolute 1e-4 of BASELINE.json's north star; the numbers are printed (pytest -s) and copied into DESIGN.md.
Layer level (one convolution against float64): tests/test_gpu_split.py::test_conv_split."""
import numpy as np
import pytest
import torch

from oracle import denoising as oden
from oracle import scoring as oscoring

pytestmark = pytest.mark.gpu
ATOL = 1e-4
RATIO = 2.0


def _abs(a, b):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max())


@pytest.mark.parametrize('arch,units', [('resnet8', 64), ('resnet16', 64), ('resnet8', 32)])
def test_scoring_nets_against_float64(gpu_ctx, arch, units):
    """filled ResNets (the bench's resnet8-u64 first) on a 512^2 N(0,1) micrograph: float64 oracle vs torch fp32 oracle vs
    the 2xf16 kernels vs the fp32-MFMA kernels"""
    from topaz_amd.model.classifier import LinearClassifier
    sd = oscoring.synthetic_resnet_sd(arch, units, 7)
    m = LinearClassifier(arch, sd)
    m.eval(); m.fill(); m.cuda()
    x = np.random.RandomState(1000).randn(512, 512).astype(np.float32)
    ref64 = oscoring.score(arch, sd, x, dtype=torch.float64)
    ref32 = oscoring.score(arch, sd, x)
    xt = torch.from_numpy(x).cuda()[None, None]
    dm = m.device_model
    before = dm.split_stats()
    y = m(xt)[0, 0].cpu().numpy()
    after = dm.split_stats()
    assert after[1] == before[1] + 1 and after[2] == before[2], 'expected one 2xf16 forward without an fp32 re-run'
    gpu_ctx.set_exact(True)
    try:
        y32 = m(xt)[0, 0].cpu().numpy()
    finally:
        gpu_ctx.set_exact(False)
    e_split, e_k32, e_t32 = _abs(y, ref64), _abs(y32, ref64), _abs(ref32, ref64)
    rms = lambda a: float(np.sqrt(np.mean((a.astype(np.float64) - ref64) ** 2)))
    print(f'{arch}-u{units} 512^2 vs float64 (logits in [{ref64.min():.1f}, {ref64.max():.1f}]):  max |2xf16| {e_split:.2e}  '
          f'|fp32 kernels| {e_k32:.2e}  |torch fp32| {e_t32:.2e};  rms {rms(y):.2e} / {rms(y32):.2e} / {rms(ref32):.2e}')
    assert e_split <= ATOL and e_k32 <= ATOL
    assert e_split <= RATIO * e_t32, (e_split, e_t32)
    assert rms(y) <= RATIO * rms(ref32)


@pytest.mark.parametrize('gain,offset', [(4.0, 2.0), (0.05, 0.0)])
def test_resnet16_u64_wide_and_narrow_inputs_against_float64(gpu_ctx, gain, offset):
    """the CLI-default detector on micrographs that are NOT N(0,1) (extract.py:234-249 scores whatever it is given): a wide
    image (x * 4 + 2: logits several times the usual range, activations up to a few hundred -- still inside f16, so no range
    scaling: s = 0 up to a 99.9 % quantile of 8) and a faint one (x * 0.05: activations towards the lo halves' absolute floor).
    Error against float64 <= 2x torch-fp32's, and <= 1e-4 absolute where the logits have the range the bar is stated for."""
    from topaz_amd.model.classifier import LinearClassifier
    sd = oscoring.synthetic_resnet_sd('resnet16', 64, 7)
    m = LinearClassifier('resnet16', sd)
    m.eval(); m.fill(); m.cuda()
    x = (np.random.RandomState(1002).randn(400, 432) * gain + offset).astype(np.float32)
    ref64 = oscoring.score('resnet16', sd, x, dtype=torch.float64)
    ref32 = oscoring.score('resnet16', sd, x)
    dm = m.device_model
    before = dm.split_stats()
    y = m(torch.from_numpy(x).cuda()[None, None])[0, 0].cpu().numpy()
    after = dm.split_stats()
    assert after[2] == before[2], 'no fp32 re-run expected'
    e, et, scale = _abs(y, ref64), _abs(ref32, ref64), float(np.abs(ref64).max())
    print(f'resnet16-u64 on x*{gain}+{offset}: |logit| up to {scale:.3g}; vs float64: 2xf16 {e:.2e}, torch fp32 {et:.2e}')
    assert e <= max(RATIO * et, 2e-6 * scale)
    if scale <= 30:
        assert e <= ATOL


@pytest.mark.parametrize('net', ['bench-nf48', 'unet-v0.2.1'])
def test_unet_against_float64(gpu_ctx, net):
    """the bench's seeded U-Net (b11 / t5, 48 filters) and the pretrained v0.2.1 on a 512 x 480 micrograph, whole image
    (Denoise._denoise, denoise.py:274-296: own mean / unbiased std, model, un-normalise)"""
    from topaz_amd.denoise import Denoise
    from topaz_amd.denoising.models import DenoiseNet
    if net == 'bench-nf48':
        sd = oden.synthetic_unet_sd(11, nf=48, base_width=11, top_width=5)
        d = Denoise(DenoiseNet('unet', sd))
    else:
        d = Denoise('unet-v0.2.1')
        sd = {k: v.numpy() for k, v in d.model.state_dict().items()}
    x = np.random.RandomState(1000).randn(512, 480).astype(np.float32)
    tsd = oden.to_torch_sd(sd)
    tsd64 = {k: v.double() for k, v in tsd.items()}
    ref64 = oden.denoise_whole('unet', tsd64, torch.from_numpy(x).double())
    ref32 = oden.denoise_whole('unet', tsd, torch.from_numpy(x))
    dm = d.model.device_model
    before = dm.split_stats()
    y = d.denoise_device(torch.from_numpy(x).cuda(), -1, 0).cpu().numpy()
    after = dm.split_stats()
    assert after[1] > before[1] and after[2] == before[2], 'expected a 2xf16 forward without an fp32 re-run'
    gpu_ctx.set_exact(True)
    try:
        y32 = d.denoise_device(torch.from_numpy(x).cuda(), -1, 0).cpu().numpy()
    finally:
        gpu_ctx.set_exact(False)
    e_split, e_k32, e_t32 = _abs(y, ref64), _abs(y32, ref64), _abs(ref32, ref64)
    print(f'{net} 512x480 vs float64 (pixels in [{ref64.min():.2f}, {ref64.max():.2f}]):  max |2xf16| {e_split:.2e}  '
          f'|fp32 kernels| {e_k32:.2e}  |torch fp32| {e_t32:.2e}')
    assert e_split <= ATOL and e_k32 <= ATOL
    assert e_split <= RATIO * max(e_t32, 1e-6), (e_split, e_t32)
