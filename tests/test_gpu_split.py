"""The 2xf16 path (topaz_amd/csrc/conv_split.h): every fp32 operand carried as two f16 halves, three exact
products accumulated in fp32 on v_mfma_f32_16x16x32_f16.  Network-level tests (whole ResNets / U-Nets against the
oracle) assert the north-star bar, 1e-4 ABSOLUTE (`_abs`); the single-layer unit tests, whose synthetic outputs reach
|y| ~ 100, use 1e-4 * (1 + |ref|) (`_err`) plus a direct comparison with a float64 evaluation showing the error is
at the fp32 level; and the f16-range fallback.  The benchmark's own nets at 4096^2: tests/test_gpu_fullsize.py."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _err(a, b):
    a = a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    b = b.detach().cpu().double().numpy() if torch.is_tensor(b) else np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((np.abs(a - b) / (1.0 + np.abs(b))).max())


def _abs(a, b):
    a = a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    b = b.detach().cpu().double().numpy() if torch.is_tensor(b) else np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a - b).max())


def _act(y, slope):
    return torch.where(y > 0, y, y * slope)


SPLIT_CASES = [
    # cin, cout, k, dil, H, W, slope
    (128, 128, 5, 4, 70, 81, 0.0),      # K5 D4 (plain epilogue of the head layer's kernel)
    (64, 128, 1, 1, 33, 47, 1.0),       # 1x1 projection
    (128, 128, 3, 4, 60, 75, 0.0),
    (128, 128, 3, 8, 70, 66, 0.0),
    (64, 128, 3, 2, 45, 50, 0.0),
    (64, 64, 3, 1, 40, 41, 0.0),
    (64, 64, 3, 2, 64, 64, 0.0),
    (64, 64, 3, 4, 37, 130, 0.25),
    (128, 256, 5, 4, 40, 49, 0.0),      # two co-groups
    (72, 100, 3, 4, 41, 43, 0.0),       # channels that are not multiples of the cell / tile sizes
]


@pytest.mark.parametrize('case', SPLIT_CASES)
def test_conv_split(gpu_ctx, case):
    from topaz_amd import runtime as rt
    cin, cout, k, dil, H, W, slope = case
    g = torch.Generator().manual_seed(31)
    x = torch.randn(cin, H, W, generator=g) * 3
    w = torch.randn(cout, cin, k, k, generator=g) / np.sqrt(cin * k * k)
    w[0] *= 1e-3                                         # per-channel weight scaling must cope with both
    w[1] *= 50
    b = torch.randn(cout, generator=g)
    ref64 = _act(F.conv2d(x[None].double(), w.double(), b.double(), dilation=dil), slope)[0]
    ref32 = _act(F.conv2d(x[None], w, b, dilation=dil), slope)[0]
    y, ovf = rt.conv_split(x, w.numpy(), b.numpy(), dil=dil, slope=slope)
    assert not ovf
    e_split, e_f32 = _err(y, ref64), _err(ref32, ref64)
    print(f'conv_split {case}: err vs float64 (scaled by 1 + |ref|): 2xf16 {e_split:.2e}, torch fp32 {e_f32:.2e}')
    assert _err(y, ref32) <= 1e-4
    # against float64 the layer is as accurate as torch's fp32 convolution of the same layer, to within a factor 1.5 (the
    # bound DESIGN.md 3.1 states; 22-bit operands, fp32 accumulation) -- not merely "not f16-level"
    assert e_split <= max(1.5 * e_f32, 1e-6), (e_split, e_f32)


def test_conv_split_residual_bn(gpu_ctx):
    from topaz_amd import runtime as rt
    g = torch.Generator().manual_seed(32)
    cin, cout, H, W, d, crop = 64, 128, 70, 60, 4, 6
    x = torch.randn(cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(cin * 9)
    b = torch.randn(cout, generator=g)
    Ho, Wo = H - 2 * d, W - 2 * d
    res = torch.randn(cout, Ho + 2 * crop, Wo + 2 * crop, generator=g)
    ps, pt = 1 + 0.1 * torch.randn(cout, generator=g), torch.randn(cout, generator=g)
    y0 = F.conv2d(x[None], w, b, dilation=d)[0] + res[:, crop:-crop, crop:-crop]
    for post in (False, True):
        ref = F.relu(y0 * ps[:, None, None] + pt[:, None, None]) if post else F.relu(y0)
        y, ovf = rt.conv_split(x, w.numpy(), b.numpy(), dil=d, slope=0.0, res=res, res_crop=crop,
                               post_scale=ps.numpy() if post else None, post_shift=pt.numpy() if post else None)
        assert not ovf and _err(y, ref) <= 1e-4


@pytest.mark.parametrize('cout', [256, 200])
def test_conv_split_fused_head(gpu_ctx, cout):
    from topaz_amd import runtime as rt
    g = torch.Generator().manual_seed(33)
    x = torch.randn(128, 50, 77, generator=g)
    w = torch.randn(cout, 128, 5, 5, generator=g) / np.sqrt(128 * 25)
    b = torch.randn(cout, generator=g)
    hw = torch.randn(cout, generator=g) / np.sqrt(cout)
    feat = F.relu(F.conv2d(x[None], w, b, dilation=4))
    ref = F.conv2d(feat, hw.view(1, cout, 1, 1), torch.tensor([0.7]))[0]
    y, _ = rt.conv_split(x, w.numpy(), b.numpy(), dil=4, slope=0.0, head_w=hw.numpy(), head_b=0.7)
    assert _err(y, ref) <= 1e-4


def test_conv_split_padding_and_overflow_flag(gpu_ctx):
    from topaz_amd import runtime as rt
    g = torch.Generator().manual_seed(34)
    x = torch.randn(64, 20, 45, generator=g)
    w = torch.randn(64, 64, 3, 3, generator=g) / 24
    ref = F.conv2d(x[None], w, None, padding=1)[0]
    y, ovf = rt.conv_split(x, w.numpy(), None, pad=1, slope=1.0)
    assert not ovf and _err(y, ref) <= 1e-4
    y, ovf = rt.conv_split(x * 1e5, w.numpy(), None, pad=1, slope=1.0)        # outputs ~1e5 > 65504
    assert ovf


def _resnet(arch, units, bn, seed=7):
    from oracle import scoring as oscoring
    from topaz_amd.model.classifier import LinearClassifier
    sd = oscoring.synthetic_resnet_sd(arch, units, seed, bn=bn)
    m = LinearClassifier(arch, sd)
    m.eval(); m.fill(); m.cuda()
    return m, sd


@pytest.mark.parametrize('arch,bn', [('resnet8', False), ('resnet8', True), ('resnet16', False)])
def test_resnet_u64_runs_split_and_matches_oracle(gpu_ctx, arch, bn):
    """A whole filled ResNet at 64 units goes down the 2xf16 path (stats say so) and agrees with the oracle and
    with the pinned fp32 kernels."""
    from oracle import scoring as oscoring
    m, sd = _resnet(arch, 64, bn)
    x = np.random.RandomState(3).randn(150, 161).astype(np.float32)
    ref = oscoring.score(arch, sd, x)
    xt = torch.from_numpy(x).cuda()[None, None]
    dm = m.device_model
    before = dm.split_stats()
    assert before[0], 'model should be eligible for the 2xf16 path'
    y = m(xt)[0, 0].cpu().numpy()
    after = dm.split_stats()
    assert after[1] == before[1] + 1 and after[2] == before[2]
    assert _abs(y, ref) <= 1e-4
    gpu_ctx.set_exact(True)
    try:
        y32 = m(xt)[0, 0].cpu().numpy()
        assert dm.split_stats()[1] == after[1]
    finally:
        gpu_ctx.set_exact(False)
    assert _abs(y32, ref) <= 1e-4 and _abs(y, y32) <= 1e-4


def test_out_of_range_activations_rerun_in_fp32(gpu_ctx):
    """Range scaling off, input scaled so that feature maps exceed 65504: the device flag trips and the image is re-run on
    the fp32 kernels -- the scores still match the oracle.  (With range scaling on -- the default -- the same image stays on
    the 2xf16 path: next test.)"""
    from oracle import scoring as oscoring
    m, sd = _resnet('resnet8', 64, False)
    x = (np.random.RandomState(4).randn(140, 150) * 3e4).astype(np.float32)
    ref = oscoring.score('resnet8', sd, x)
    dm = m.device_model
    before = dm.split_stats()
    gpu_ctx.set_range(False)
    try:
        y = m(torch.from_numpy(x).cuda()[None, None])[0, 0].cpu().numpy()
    finally:
        gpu_ctx.set_range(True)
    after = dm.split_stats()
    assert after[2] == before[2] + 1, 'expected the fp32 re-run'
    assert np.isfinite(y).all()
    # logits scale with the input here (~1e5): compare relative to their magnitude
    assert float(np.abs(y - ref).max()) <= 1e-5 * float(np.abs(ref).max())
    gpu_ctx.set_exact(True)
    try:
        y32 = m(torch.from_numpy(x).cuda()[None, None])[0, 0].cpu().numpy()
    finally:
        gpu_ctx.set_exact(False)
    assert np.array_equal(y, y32), 'the re-run must be the fp32 path itself'


@pytest.mark.parametrize('net', ['resnet8_u64', 'resnet8_u64_bn', 'resnet16_u64', 'resnet8_u32_pretrained', 'conv63_u32_bn'])
def test_raw_count_micrograph_stays_on_the_2xf16_path(gpu_ctx, net):
    """`topaz extract` does not normalise what it scores (extract.py:234-249): an un-normalised, detector-count micrograph
    (uint16 range, mean 3000) would overflow f16 in the stem.  The pass runs on x * 2^-s with every bias scaled by 2^-s and
    multiplies the logits back (the network is homogeneous in (input, biases); powers of two are exact): NO fp32 re-run, logits
    as close to a float64 evaluation as torch's own fp32 evaluation is (<= 2x its error), and equal to the exact-fp32 kernels'
    within the same bound."""
    from oracle import scoring as oscoring
    if net == 'resnet8_u32_pretrained':
        from topaz_amd.model.factory import load_model
        m = load_model('resnet8_u32')
        m.eval(); m.fill(); m.cuda()
        arch, sd = 'resnet8', {k: v.numpy() for k, v in m.state_dict().items()}
    elif net == 'conv63_u32_bn':
        from topaz_amd.model.classifier import LinearClassifier
        arch, sd = 'conv63', oscoring.synthetic_basic_sd((7, 5, 5, 5), 32, 5, bn=True)
        m = LinearClassifier(arch, sd)
        m.eval(); m.fill(); m.cuda()
    else:
        arch = net.split('_')[0]
        m, sd = _resnet(arch, 64, net.endswith('_bn'))
    rs = np.random.RandomState(21)
    x = np.clip(np.round(rs.randn(200, 232) * 400 + 3000), 0, 65535).astype(np.float32)
    x[17, 40] = 65535.0                                   # a hot pixel
    ref64 = oscoring.score(arch, sd, x, dtype=torch.float64)
    ref32 = oscoring.score(arch, sd, x)
    dm = m.device_model
    before = dm.split_stats()
    y = m(torch.from_numpy(x).cuda()[None, None])[0, 0].cpu().numpy()
    after = dm.split_stats()
    assert after[1] == before[1] + 1 and after[2] == before[2], f'fp32 re-runs: {after[2] - before[2]}'
    e_ours = float(np.abs(y - ref64).max())
    e_torch = float(np.abs(ref32.astype(np.float64) - ref64).max())
    scale = float(np.abs(ref64).max())
    print(f'{net}: |logit| up to {scale:.3g}; error vs float64: 2xf16 + range scaling {e_ours:.3g}, torch fp32 {e_torch:.3g}')
    assert e_ours <= max(2.0 * e_torch, 2e-6 * scale)
    gpu_ctx.set_exact(True)
    try:
        y32 = m(torch.from_numpy(x).cuda()[None, None])[0, 0].cpu().numpy()
    finally:
        gpu_ctx.set_exact(False)
    assert float(np.abs(y - y32).max()) <= max(3.0 * e_torch, 3e-6 * scale)
    # an image inside +-32 is not touched: identical bits with the switch off
    xn = ((x - 3000) / 400).astype(np.float32)
    xn[17, 40] = 31.0
    a = m(torch.from_numpy(xn).cuda()[None, None])[0, 0].cpu().numpy()
    gpu_ctx.set_range(False)
    try:
        b = m(torch.from_numpy(xn).cuda()[None, None])[0, 0].cpu().numpy()
    finally:
        gpu_ctx.set_range(True)
    assert np.array_equal(a, b)


@pytest.mark.parametrize('hot', [3.0e4, 1.0e6])
def test_hot_pixels_do_not_starve_the_rest_of_precision(gpu_ctx, hot):
    """The range exponent follows the BULK of the image (99.9 % quantile), not its maximum: a normalised micrograph with a few
    hot pixels keeps s = 0, so every logit outside the hot pixels' receptive fields is the one the clean image gives, to the
    bit or within 1e-4 of the oracle (had the maximum decided, the N(0,1) pixels would have been scaled towards the f16
    subnormals and lost their lo halves without any flag).  3e4 still fits f16 (2xf16 path, maybe an fp32 re-run when an
    activation overflows), 1e6 does not (re-run): right either way."""
    from oracle import scoring as oscoring
    m, sd = _resnet('resnet8', 64, False)
    rs = np.random.RandomState(12)
    x = rs.randn(420, 400).astype(np.float32)
    xh = x.copy()
    for (py, px) in ((5, 7), (210, 200), (215, 203), (400, 30)):
        xh[py, px] = hot
    ref = oscoring.score('resnet8', sd, xh)
    y = m(torch.from_numpy(xh).cuda()[None, None])[0, 0].cpu().numpy()
    assert np.isfinite(y).all()
    far = np.ones_like(x, dtype=bool)                      # pixels whose 71-pixel receptive field holds no hot pixel
    for (py, px) in ((5, 7), (210, 200), (215, 203), (400, 30)):
        far[max(0, py - 36):py + 37, max(0, px - 36):px + 37] = False
    assert far.sum() > 50000
    assert float(np.abs(y - ref)[far].max()) <= 1e-4
    scale = float(np.abs(ref).max())
    assert float(np.abs(y - ref).max()) <= max(1e-4, 1e-5 * scale)
    clean = m(torch.from_numpy(x).cuda()[None, None])[0, 0].cpu().numpy()
    assert float(np.abs(y - clean)[far].max()) <= 2e-4


@pytest.mark.parametrize('shape', [(256, 320), (253, 190), (150, 140)])
def test_unet_denoise_on_split_path_matches_oracle(gpu_ctx, shape):
    """The 48-filter U-Net goes down the 2xf16 path: encoder / decoder 3x3 convs, the per-parity decoder kernels
    where the skip is exactly 2x the upsampled tensor (all levels for 256x320), fp32 kernels with on-device
    format conversion where it is not (odd sizes), dec1.2 storing fp32 for the 1-channel last conv."""
    from oracle import denoising as oden
    from topaz_amd.denoise import Denoise
    from topaz_amd.denoising.models import DenoiseNet
    sd = oden.synthetic_unet_sd(11, nf=48, base_width=11, top_width=5)
    dn = Denoise(DenoiseNet('unet', sd))
    x = np.random.RandomState(8).randn(*shape).astype(np.float32) * 3 + 1
    ref = oden.denoise('unet', sd, x)
    dm = dn.model.device_model
    before = dm.split_stats()
    assert before[0], 'the U-Net should be eligible for the 2xf16 path'
    y = dn.denoise_device(torch.from_numpy(x).cuda(), -1, 0).cpu().numpy()
    after = dm.split_stats()
    assert after[1] == before[1] + 1 and after[2] == before[2]
    assert _abs(y, ref) <= 1e-4
    gpu_ctx.set_exact(True)
    try:
        y32 = dn.denoise_device(torch.from_numpy(x).cuda(), -1, 0).cpu().numpy()
    finally:
        gpu_ctx.set_exact(False)
    assert _abs(y32, ref) <= 1e-4 and _abs(y, y32) <= 1e-4


def test_unet_patched_denoise_split_matches_oracle(gpu_ctx):
    from oracle import denoising as oden
    from topaz_amd.denoise import Denoise
    from topaz_amd.denoising.models import DenoiseNet
    sd = oden.synthetic_unet_sd(12, nf=48, base_width=11, top_width=5)
    dn = Denoise(DenoiseNet('unet', sd))
    x = np.random.RandomState(9).randn(300, 280).astype(np.float32)
    ref = oden.denoise('unet', sd, x, patch_size=128, padding=64)
    y = dn.denoise_device(torch.from_numpy(x).cuda(), 128, 64).cpu().numpy()
    assert _abs(y, ref) <= 1e-4
    assert dn.model.device_model.split_stats()[1] >= 1


def test_unet3d_on_split_path_matches_oracle(gpu_ctx):
    """UDenoiseNet3D (48 filters) on the plane-stacked 2xf16 kernels: 3-D convs as 2-D tiles over stacked planes,
    8-parity decoder kernels, 3-D split max-pool, fp32 only in the two stems and the 1-channel last conv."""
    from oracle import denoising as oden
    from topaz_amd.denoise import Denoise3D
    from topaz_amd.denoising.models import DenoiseNet
    sd = oden.synthetic_unet_sd(13, nf=48, base_width=7, top_width=3, dims=3)
    d3 = Denoise3D(DenoiseNet('unet-3d', sd))
    x = (np.random.RandomState(10).randn(64, 64, 96) * 2 + 0.5).astype(np.float32)
    ref = oden.denoise_whole('unet-3d', oden.to_torch_sd(sd), torch.from_numpy(x))
    dm = d3.model.device_model
    before = dm.split_stats()
    assert before[0], 'the 3-D U-Net should be eligible for the 2xf16 path'
    y = dm.denoise_3d(torch.from_numpy(x).cuda(), -1, 0).cpu().numpy()
    after = dm.split_stats()
    assert after[1] == before[1] + 1 and after[2] == before[2]
    assert _abs(y, ref) <= 1e-4
    gpu_ctx.set_exact(True)
    try:
        y32 = dm.denoise_3d(torch.from_numpy(x).cuda(), -1, 0).cpu().numpy()
    finally:
        gpu_ctx.set_exact(False)
    assert _abs(y32, ref) <= 1e-4 and _abs(y, y32) <= 1e-4


@pytest.mark.parametrize('nf,bw,tw', [(24, 7, 3), (40, 11, 5)])
def test_unet_odd_widths_mixed_paths(gpu_ctx, nf, bw, tw):
    """Filter counts without a full set of 2xf16 instantiations: layers that have a kernel run split, the others
    fp32, with on-device format conversion between them (cells of 8 channels, zero-padded) -- results still match."""
    from oracle import denoising as oden
    from topaz_amd.denoise import Denoise
    from topaz_amd.denoising.models import DenoiseNet
    sd = oden.synthetic_unet_sd(30 + nf, nf=nf, base_width=bw, top_width=tw)
    dn = Denoise(DenoiseNet('unet', sd))
    x = np.random.RandomState(nf).randn(192, 160).astype(np.float32)
    ref = oden.denoise('unet', sd, x)
    y = dn.denoise_device(torch.from_numpy(x).cuda(), -1, 0).cpu().numpy()
    assert _abs(y, ref) <= 1e-4


def test_conv_split_random_shapes_vs_fp32_kernels(gpu_ctx):
    """Randomised sweep: the 2xf16 kernel against the fp32 MFMA kernel of the same layer (both through the C-ABI)
    over kernel sizes / dilations / paddings / odd channel counts and image sizes around the tile boundaries."""
    from topaz_amd import runtime as rt
    rs = np.random.RandomState(77)
    combos = [(3, 1, 96), (3, 1, 48), (3, 1, 64), (3, 2, 64), (3, 4, 64), (3, 4, 128), (3, 8, 128), (3, 2, 128),
              (5, 4, 128), (1, 1, 128), (1, 1, 64), (3, 1, 32), (3, 2, 32), (3, 4, 32), (5, 2, 32), (5, 4, 32), (5, 8, 32),
              (2, 1, 96), (2, 1, 64), (5, 1, 32)]
    for case in range(40):
        k, dil, cout_max = combos[rs.randint(len(combos))]
        cout = int(cout_max if rs.rand() < 0.5 else rs.randint(max(cout_max // 2 + 1, 1), cout_max + 1))
        cin = int(rs.choice([8, 16, 24, 40, 48, 64, 72, 96, 128]))
        pad = int(rs.choice([0, k // 2, 1])) if k > 1 else 0
        span = dil * (k - 1)
        H = span + 1 - 2 * pad + int(rs.randint(1, 70))
        W = span + 1 - 2 * pad + int(rs.randint(1, 90))
        H, W = max(H, 1), max(W, 1)
        x = torch.from_numpy(rs.randn(cin, H, W).astype(np.float32))
        w = (rs.randn(cout, cin, k, k) / np.sqrt(cin * k * k)).astype(np.float32)
        b = rs.randn(cout).astype(np.float32)
        slope = float(rs.choice([0.0, 0.1, 1.0]))
        y32 = rt.conv(x, w, b, dil=dil, pad=pad, slope=slope)
        ys, ovf = rt.conv_split(x, w, b, dil=dil, pad=pad, slope=slope)
        assert not ovf
        assert _err(ys, y32) <= 1e-4, (case, k, dil, cin, cout, pad, H, W, _err(ys, y32))


def test_split_path_is_bitwise_reproducible(gpu_ctx):
    """no atomics in any accumulation: repeated runs of the same image give bit-identical logits / pixels"""
    from tools import synth_weights as sw
    from topaz_amd.denoise import Denoise
    from topaz_amd.denoising.models import DenoiseNet
    m, _ = sw.hip_resnet('resnet8', 64, 7)
    x = torch.from_numpy(np.random.RandomState(5).randn(700, 650).astype(np.float32)).cuda()
    y0 = m(x[None, None])[0, 0].clone()
    for _ in range(3):
        assert torch.equal(y0, m(x[None, None])[0, 0])
    dn = Denoise(DenoiseNet('unet', sw.unet_sd(11)))
    z0 = dn.denoise_device(x, 256, 64).clone()
    for _ in range(2):
        assert torch.equal(z0, dn.denoise_device(x, 256, 64))


@pytest.mark.parametrize('wgs', [8, 24])
@pytest.mark.parametrize('case', [
    # cin, cout, k, dil, H, W, slope, residual
    (128, 128, 3, 8, 150, 170, 0.0, True),      # 8 chunks (even): the input buffers keep their parity from tile to tile
    (128, 128, 3, 4, 120, 200, 0.0, False),     # two steps per stage
    (64, 64, 3, 2, 90, 260, 0.0, False),        # 4-wave tile, two workgroups per CU
    (64, 64, 3, 4, 130, 141, 0.25, True),
    (72, 100, 3, 4, 101, 143, 0.0, False),      # 9 cells: odd number of chunks (parity flips every tile), short last chunk
    (24, 48, 3, 1, 77, 300, 0.1, False),        # 3 cells -> 2 chunks, second one short
    (64, 32, 5, 1, 64, 200, 0.1, False),        # 5x5, two steps per stage
    (96, 96, 3, 1, 60, 333, 0.1, False),
])
def test_persistent_workgroups_are_bit_identical(gpu_ctx, case, wgs):
    """Persistent workgroups (conv_split.h MODE 4): a handful of workgroups walk all tiles of the layer, each fetching the
    first chunk / weight stage of its next tile during the last chunk of the current one.  The result must not differ by one
    bit from the one-tile-per-workgroup launch, for even and odd chunk / stage counts (buffer parities), short last chunks,
    multi-step stages, residual epilogues and images whose last tiles overhang."""
    from topaz_amd import runtime as rt
    cin, cout, k, dil, H, W, slope, with_res = case
    g = torch.Generator().manual_seed(77)
    x = torch.randn(cin, H, W, generator=g) * 2
    w = torch.randn(cout, cin, k, k, generator=g) / np.sqrt(cin * k * k)
    b = torch.randn(cout, generator=g)
    Ho, Wo = H - dil * (k - 1), W - dil * (k - 1)
    res = torch.randn(cout, Ho + 4, Wo + 4, generator=g) if with_res else None
    kw = dict(dil=dil, slope=slope, res=res, res_crop=2 if with_res else 0)
    gpu_ctx.set_persist(0)
    try:
        y0, ovf0 = rt.conv_split(x, w.numpy(), b.numpy(), **kw)
        gpu_ctx.set_persist(2, wgs)
        y1, ovf1 = rt.conv_split(x, w.numpy(), b.numpy(), **kw)
    finally:
        gpu_ctx.set_persist(1, 0)
    ref = F.conv2d(x[None], w, b, dilation=dil)[0]
    if with_res:
        ref = ref + res[:, 2:-2, 2:-2]
    ref = _act(ref, slope)
    assert not ovf0 and not ovf1
    assert _err(y0, ref) <= 1e-4
    assert torch.equal(y0, y1), float((y0 - y1).abs().max())


def test_sustained_mfma_probe_reports_a_plausible_rate(gpu_ctx):
    """tpz_prof_mfma_sustained (csrc/diag.hip): the register-resident MFMA loop bench.py quotes the dominant kernel against.
    Zero operands run at (nearly) the nominal dense rate; full-entropy operands are held below it by the power management."""
    tf_rand, clk_rand = gpu_ctx.mfma_sustained(60, False)
    tf_zero, clk_zero = gpu_ctx.mfma_sustained(60, True)
    print(f'sustained v_mfma_f32_16x16x32_f16: {tf_rand:.0f} TFLOP/s on random operands, {tf_zero:.0f} on zeros; '
          f'clock ratio {clk_rand / clk_zero:.3f}')
    assert 500 < tf_rand <= tf_zero * 1.10 and 1000 < tf_zero < 2700, (tf_rand, tf_zero)      # (wide: boxes differ, 60-ms loops)
    assert clk_rand > 0 and clk_zero > 0
