"""U-Net / FCNN / affine denoising on the MI355X against the reference's golden outputs and the
CPU oracle.  Tolerance: 1e-4 ABSOLUTE on the denoised pixels as returned (BASELINE.json north_star), also where the
inputs are x*3+10 or x*2+5."""
import numpy as np
import pytest
import torch

from conftest import golden_sd, load_golden
from oracle import denoising as oden

pytestmark = pytest.mark.gpu
ATOL = 1e-4


def _err(a, b):
    return np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)).max()


@pytest.mark.parametrize('name', ['unet-v0.2.1', 'unet-small', 'fcnn', 'affine'])
def test_pretrained_vs_reference_golden(gpu_ctx, name):
    from topaz_amd.denoise import Denoise
    z = load_golden('denoise2d_pretrained')
    d = Denoise(name)
    x = z['x']
    assert _err(d.denoise(x, patch_size=-1), z[f'{name}:whole']) <= ATOL
    assert _err(d.denoise(x, patch_size=64, padding=24), z[f'{name}:p64_24']) <= ATOL


def test_denoise_image_variants_vs_reference_golden(gpu_ctx):
    from topaz_amd.denoise import Denoise, denoise_image
    from topaz_amd.filters import GaussianDenoise
    z = load_golden('denoise2d_pretrained')
    x = z['x']
    d = Denoise('unet-v0.2.1')
    assert _err(denoise_image(x.copy(), [d], patch_size=96, padding=16), z['image:unet-v0.2.1:p96_16']) <= ATOL
    assert _err(denoise_image(x.copy(), [d], patch_size=-1, normalize=True), z['image:unet-v0.2.1:norm']) <= ATOL
    g = GaussianDenoise(1.2)
    assert _err(g.apply(x), z['gaus1.2:apply']) <= ATOL
    assert _err(denoise_image(x.copy(), [Denoise('unet-small')], gaus=g), z['image:unet-small:gaus1.2']) <= ATOL
    assert _err(denoise_image(x.copy(), [Denoise('affine')], cutoff=1.5), z['image:affine:cutoff']) <= ATOL


def test_seeded_v022_arch_vs_reference_golden(gpu_ctx):
    from topaz_amd.denoise import Denoise
    from topaz_amd.denoising.models import DenoiseNet
    z = load_golden('denoise2d_unet_b11t5_nf16')
    d = Denoise(DenoiseNet('unet', golden_sd(z)))
    assert _err(d.denoise(z['x'], -1), z['whole']) <= ATOL
    assert _err(d.denoise(z['x'], 48, 20), z['p48_20']) <= ATOL


def test_unet_v022_nf48_vs_oracle(gpu_ctx):
    """the CLI-default architecture (base 11, top 5, nf 48; blob missing upstream), seeded weights,
    odd pooled sizes (381 -> 190 -> 95 -> 47 -> 23 -> 11) and default-style patching"""
    from topaz_amd.denoise import Denoise
    from topaz_amd.denoising.models import DenoiseNet
    sd = oden.synthetic_unet_sd(11, nf=48, base_width=11, top_width=5)
    x = np.random.RandomState(1002).randn(381, 400).astype(np.float32)
    d = Denoise(DenoiseNet('unet', sd))
    assert _err(d.denoise(x, -1), oden.denoise('unet', sd, x, -1)) <= ATOL
    assert _err(d.denoise(x, 128, 60), oden.denoise('unet', sd, x, 128, 60)) <= ATOL


def test_denoise3d_vs_reference_golden(gpu_ctx):
    from topaz_amd.denoise import Denoise3D
    from topaz_amd.denoising.models import DenoiseNet
    z = load_golden('denoise3d_unet3d_nf8')
    d = Denoise3D(DenoiseNet('unet-3d', golden_sd(z)))
    assert _err(d.denoise(z['tomo'], 32, 16, verbose=False), z['p32_16']) <= ATOL
    assert _err(d.denoise(z['small'], -1, verbose=False), z['small_whole']) <= ATOL


def test_user_model_pickle(gpu_ctx):
    import os
    from conftest import GOLDEN
    from topaz_amd.denoise import Denoise
    z = load_golden('denoise2d_unet_b11t5_nf16')
    d = Denoise(os.path.join(GOLDEN, 'user_model_unet_b11t5_nf16.sav'))
    assert _err(d.denoise(z['x'], 48, 20), z['p48_20']) <= ATOL


@pytest.mark.parametrize('tag', ['unet2_nf12', 'unet3'])
def test_user_trainable_archs_unet2_unet3_pickles(gpu_ctx, tag):
    """`topaz denoise --arch unet2 | unet3` models (denoising/models.py:247-449) loaded from their full-module pickles by class
    name: UDenoiseNet2 drops the skip connections into dec2 / dec1, UDenoiseNet3 returns x - dec1(h) (the residual is added in
    the last conv's epilogue).  Against the reference's own outputs, default path and exact-fp32 path."""
    import os
    from conftest import GOLDEN
    from topaz_amd.denoise import Denoise
    import warnings
    z = load_golden(f'denoise2d_{tag}')
    with warnings.catch_warnings():
        # widths that are not multiples of 16 (nf = 12: sources of 12 + 12 and 24 + 1 channels) are loaded zero-padded to the
        # next multiple (rt_load.hip widen_program): every layer on the 2xf16 path, no mixed-program warning
        warnings.simplefilter('error')
        d = Denoise(os.path.join(GOLDEN, f'user_model_{tag}.sav'))
    n_conv, n_split, off = d.model.device_model.split_layers()
    assert n_split == n_conv, off
    assert d.model.kind == tag.split('_')[0]
    for exact in (False, True):
        gpu_ctx.set_exact(exact)
        try:
            assert _err(d.denoise(z['x'], -1), z['whole']) <= ATOL
            assert _err(d.denoise(z['x'], 96, 24), z['p96_24']) <= ATOL
        finally:
            gpu_ctx.set_exact(False)


def test_unet3d_nf48_tile_vs_oracle(gpu_ctx):
    """the pretrained 3-D architecture (nf 48, base 7; blob missing upstream) with seeded weights on
    one 64^3 tile through Denoise3D._denoise, and a 2x2x1 tile grid with 32/16 patches"""
    from topaz_amd.denoise import Denoise3D
    from topaz_amd.denoising.models import DenoiseNet
    sd = oden.synthetic_unet_sd(13, nf=48, base_width=7, top_width=3, dims=3)
    d = Denoise3D(DenoiseNet('unet-3d', sd))
    v = np.random.RandomState(2000).randn(64, 64, 64).astype(np.float32)
    ref = oden.denoise3d(sd, v, -1, 0)
    assert _err(d.denoise(v, -1, verbose=False), ref) <= ATOL
    t = np.random.RandomState(2001).randn(30, 40, 50).astype(np.float32) * 2 + 3
    ref = oden.denoise3d(sd, t, 32, 16)
    assert _err(d.denoise(t, 32, 16, verbose=False), ref) <= ATOL


def test_unet3d_odd_sizes_whole_volume(gpu_ctx):
    """odd extents in every axis through the whole 3-D U-Net (floor pooling five times, nearest upsampling back onto odd
    skip tensors): the in-plane max-pool fused into the plane-stacked convs, the z-pair kernel and the fused / per-parity
    decoder paths against the oracle, default path and fp32 kernels"""
    from topaz_amd.denoise import Denoise3D
    from topaz_amd.denoising.models import DenoiseNet
    sd = oden.synthetic_unet_sd(13, nf=48, base_width=7, top_width=3, dims=3)
    d = Denoise3D(DenoiseNet('unet-3d', sd))
    for shape, seed in (((41, 50, 37), 5), ((33, 35, 63), 6)):
        v = (np.random.RandomState(seed).randn(*shape) * 1.5 - 0.5).astype(np.float32)
        ref = oden.denoise3d(sd, v, -1, 0)
        assert _err(d.denoise(v, -1, verbose=False), ref) <= ATOL, shape
        gpu_ctx.set_exact(True)
        try:
            assert _err(d.denoise(v, -1, verbose=False), ref) <= ATOL, shape
        finally:
            gpu_ctx.set_exact(False)


def test_full_size_4096_default_patching_vs_oracle_patch(gpu_ctx):
    """BASELINE size with the CLI-default patching (-s 1024 -p 500): the output inside one patch's centre must
    equal the oracle's _denoise of that patch crop alone (patches are independent, denoise.py:307-322)."""
    from topaz_amd.denoise import Denoise
    x = (np.random.RandomState(1003).randn(4096, 4096) * 2 + 5).astype(np.float32)
    d = Denoise('unet-v0.2.1')
    y = d.denoise(x, patch_size=1024, padding=500)
    assert y.shape == x.shape
    sd = {k: v.numpy() for k, v in d.model.state_dict().items()}
    import torch
    for (i, j) in ((0, 0), (3072, 1024)):
        si, ei, sj, ej = max(0, i - 500), min(4096, i + 1524), max(0, j - 500), min(4096, j + 1524)
        ref = oden.denoise_whole('unet', oden.to_torch_sd(sd), torch.from_numpy(x[si:ei, sj:ej].copy()))
        ref = ref[i - si:i - si + 1024, j - sj:j - sj + 1024]
        assert _err(y[i:i + 1024, j:j + 1024], ref) <= ATOL, (i, j)


def test_full_size_tomogram_c5_tiles_vs_oracle(gpu_ctx):
    """BASELINE config 5: 512x512x256 tomogram, unet-3d architecture (nf 48, base 7; seeded), 96/48 tiles
    (108 tiles of 192^3).  Two tiles are recomputed with the oracle from the reference's tiling rule
    (datasets.py:426-468 zero-filled crop, global (x-mu)/std, per-tile _denoise, *std+mu)."""
    import torch
    from topaz_amd.denoise import Denoise3D
    from topaz_amd.denoising.models import DenoiseNet
    sd = oden.synthetic_unet_sd(13, nf=48, base_width=7, top_width=3, dims=3)
    tomo = np.random.RandomState(2000).randn(256, 512, 512).astype(np.float32)
    d = Denoise3D(DenoiseNet('unet-3d', sd))
    y = d.denoise(tomo, 96, 48, verbose=False)
    assert y.shape == tomo.shape and np.isfinite(y).all()
    mu, std = tomo.mean(), tomo.std()
    tsd = oden.to_torch_sd(sd)
    for (i, j, k) in ((0, 0, 0), (192, 480, 288)):
        x = np.zeros((192, 192, 192), dtype=np.float32)
        si, ei = max(0, i - 48), min(256, i + 144)
        sj, ej = max(0, j - 48), min(512, j + 144)
        sk, ek = max(0, k - 48), min(512, k + 144)
        x[48 - i + si:48 - i + ei, 48 - j + sj:48 - j + ej, 48 - k + sk:48 - k + ek] = tomo[si:ei, sj:ej, sk:ek]
        out = oden.denoise_whole('unet-3d', tsd, (torch.from_numpy(x) - mu) / std) * std + mu
        pz, py, px = min(96, 256 - i), min(96, 512 - j), min(96, 512 - k)
        ref = out[48:48 + pz, 48:48 + py, 48:48 + px]
        assert _err(y[i:i + pz, j:j + py, k:k + px], ref) <= ATOL, (i, j, k)


@pytest.mark.parametrize('exact', [False, True], ids=['2xf16', 'exact_fp32'])
@pytest.mark.parametrize('case', ['bench_net', 'pretrained', 'small', 'fcnn', 'tight_padding'])
def test_patch_windows_are_bit_identical(gpu_ctx, case, exact):
    """a patch keeps only its centre, so every layer computes only the rectangle the kept pixels depend on
    (rt_exec.hip need_regions): the output must not change by one bit against computing every tensor in full --
    corner, edge and interior patches, odd sizes (pooled sizes not divisible by two), windows clipped at the borders,
    padding smaller than the receptive field (nothing to save: the windows must then cover everything).
    exact_fp32: the same on the fp32-MFMA / direct kernels (tpz_ctx_set_exact), whose launches take the same windows"""
    from topaz_amd.denoise import Denoise
    from topaz_amd.denoising.models import DenoiseNet
    if case == 'bench_net':
        d = Denoise(DenoiseNet('unet', oden.synthetic_unet_sd(11, nf=48, base_width=11, top_width=5)))
        shape, patch, pad = (1500, 1330), 400, 300
    elif case == 'pretrained':
        d = Denoise('unet-v0.2.1')
        shape, patch, pad = (1111, 1201), 256, 280
    elif case == 'small':
        d = Denoise('unet-small')
        shape, patch, pad = (700, 900), 200, 120
    elif case == 'fcnn':
        d = Denoise('fcnn')
        shape, patch, pad = (500, 640), 128, 64
    else:
        d = Denoise('unet-v0.2.1')
        shape, patch, pad = (600, 700), 192, 40
    x = (np.random.RandomState(77).randn(*shape) * 3 + 1).astype(np.float32)
    try:
        gpu_ctx.set_exact(exact)
        gpu_ctx.set_roi(False)
        full = d.denoise(x, patch, pad)
        gpu_ctx.set_roi(True)
        win = d.denoise(x, patch, pad)
    finally:
        gpu_ctx.set_roi(True)
        gpu_ctx.set_exact(False)
    assert np.isfinite(full).all()
    assert np.array_equal(full, win)
    eligible, split_runs, fp32_reruns = d.model.device_model.split_stats()
    assert fp32_reruns == 0            # (a window never raises the overflow flag on pixels nobody computed)
    if exact:
        assert split_runs == 0
    elif case != 'fcnn':
        assert eligible and split_runs >= 2


def test_patch_windows_random_geometries(gpu_ctx):
    """seeded sweep of image sizes, patch sizes and paddings (odd sizes, paddings below and above the receptive field,
    patches that do not divide the image): windowed and full computation agree bit for bit"""
    from topaz_amd.denoise import Denoise
    rs = np.random.RandomState(4242)
    nets = [Denoise('unet-v0.2.1'), Denoise('unet-small')]
    try:
        for it in range(10):
            d = nets[it % 2]
            H, W = int(rs.randint(180, 900)), int(rs.randint(180, 900))
            patch, pad = int(rs.randint(48, 320)), int(rs.randint(8, 260))
            x = (rs.randn(H, W) * 2 - 0.5).astype(np.float32)
            gpu_ctx.set_exact(it >= 6)              # the last four on the fp32 kernels
            gpu_ctx.set_roi(False)
            full = d.denoise(x, patch, pad)
            gpu_ctx.set_roi(True)
            win = d.denoise(x, patch, pad)
            assert np.array_equal(full, win), (it, H, W, patch, pad)
    finally:
        gpu_ctx.set_roi(True)
        gpu_ctx.set_exact(False)


def test_edge_cases(gpu_ctx):
    from topaz_amd._lib import TopazHipError
    from topaz_amd.denoise import Denoise
    d = Denoise('unet-v0.2.1')
    sd = {k: v.numpy() for k, v in d.model.state_dict().items()}
    # the smallest image the 5-level U-Net accepts (32 -> 16 -> 8 -> 4 -> 2 -> 1) and a ragged one
    for shape in ((32, 32), (33, 47)):
        x = np.random.RandomState(shape[1]).randn(*shape).astype(np.float32)
        assert _err(d.denoise(x, -1), oden.denoise('unet', sd, x, -1)) <= ATOL
    # too small to pool five times: the reference dies inside max_pool2d; here a clear error
    with pytest.raises(TopazHipError, match='too small'):
        d.denoise(np.random.randn(20, 40).astype(np.float32), -1)
    # constant image: std = 0 -> the reference returns NaN everywhere; so do we
    y = d.denoise(np.full((40, 40), 3.0, dtype=np.float32), -1)
    assert np.isnan(y).all()


def test_tomogram_tiles_sharded_like_ranks_would(gpu_ctx):
    """tpz_denoise_3d_shard: the tiles of one tomogram dealt round-robin to n shards (what n ranks do, each followed by one
    RCCL reduce onto rank 0) -- the shards' volumes are disjoint and their sum is bit-identical to the unsharded result"""
    from topaz_amd.denoise import Denoise3D
    from topaz_amd.denoising.models import DenoiseNet
    z = load_golden('denoise3d_unet3d_nf8')
    d = Denoise3D(DenoiseNet('unet-3d', golden_sd(z)))
    tomo = torch.from_numpy(z['tomo']).cuda()
    dm = d.model.device_model
    whole = dm.denoise_3d(tomo, 32, 16)
    for n in (2, 3):
        parts = [dm.denoise_3d(tomo, 32, 16, shard=k, n_shards=n) for k in range(n)]
        covered = sum((p != 0).to(torch.int32) for p in parts)
        assert int(covered.max()) <= 1                       # no voxel written by two shards
        assert torch.equal(sum(parts), whole)
    assert np.abs(whole.cpu().numpy() - z['p32_16']).max() <= ATOL
    with pytest.raises(Exception, match='cannot be sharded'):
        dm.denoise_3d(tomo, -1, 0, shard=0, n_shards=2)


def test_batch_is_cut_to_the_memory_it_may_take(gpu_ctx):
    """every image of a batch has a workspace of its own (rt_denoise.hip batch_that_fits): with no memory to spend the pass falls
    back to single patches on the lanes -- the launch count of tpz_ctx_set_batch(0), the same bits -- instead of failing"""
    from topaz_amd.denoise import Denoise
    d = Denoise('unet-small')
    x = (np.random.RandomState(79).randn(700, 900) * 3 + 1).astype(np.float32)

    def run():
        n0 = gpu_ctx.launches()
        y = d.denoise(x, 200, 120)
        return y, gpu_ctx.launches() - n0
    try:
        run()                                   # (first use: anything issued once per model stays out of the counts)
        y8, n8 = run()
        gpu_ctx.set_batch_memory(1)
        y_none, n_none = run()
        gpu_ctx.set_batch_memory(1 << 42)
        y_all, n_all = run()
        gpu_ctx.set_batch_memory(0)
        gpu_ctx.set_batch(0)
        y0, n0 = run()
    finally:
        gpu_ctx.set_batch(8)
        gpu_ctx.set_batch_memory(0)
    assert n_none == n0 > n8 == n_all, (n8, n_none, n_all, n0)
    assert np.array_equal(y8, y_none) and np.array_equal(y8, y_all) and np.array_equal(y8, y0)


@pytest.mark.parametrize('case', ['golden_nf8', 'nf48_ragged', 'nf48_odd_levels', 'nf48_96_48'])
def test_tile_windows_3d_are_bit_identical(gpu_ctx, case):
    """Denoise3D keeps the centre patch^3 of every (patch + 2*padding)^3 tile (denoise.py:340-377; 1/8 of the tile at the CLI's
    96 / 48): on the 2xf16 path every layer computes only the box the kept voxels depend on (need_regions with boxes, the
    plane-stacked launches' wz0 / Dout).  Identical bits to computing every tensor in full; tiles clipped at the volume's
    border, odd extents; no overflow re-run caused by voxels nobody computed; and the windows do save work (launch FLOP
    accounted by the library).  nf48_odd_levels: 80^3 tiles pool to 5^3 and 2^3, the decoder at that level is not an exact 2x
    upsampling and the layer has no windowed kernel of its own -- the program must then be left whole, not half-windowed."""
    from topaz_amd.denoise import Denoise3D
    from topaz_amd.denoising.models import DenoiseNet
    if case == 'golden_nf8':
        z = load_golden('denoise3d_unet3d_nf8')
        d = Denoise3D(DenoiseNet('unet-3d', golden_sd(z)))
        vol, patch, pad = z['tomo'], 32, 16
    else:
        d = Denoise3D(DenoiseNet('unet-3d', oden.synthetic_unet_sd(13, nf=48, base_width=7, top_width=3, dims=3)))
        if case == 'nf48_ragged':
            vol, patch, pad = (np.random.RandomState(31).randn(50, 77, 90) * 2 + 1).astype(np.float32), 32, 16
        elif case == 'nf48_odd_levels':
            vol, patch, pad = (np.random.RandomState(33).randn(45, 50, 85) * 2 + 1).astype(np.float32), 40, 20
        else:
            vol, patch, pad = np.random.RandomState(32).randn(100, 200, 120).astype(np.float32), 96, 48
    dm = d.model.device_model
    t = torch.from_numpy(vol).cuda()
    try:
        gpu_ctx.set_roi(False)
        gpu_ctx.prof_enable(1); gpu_ctx.prof_reset()
        full = dm.denoise_3d(t, patch, pad).cpu().numpy()
        fl_full = gpu_ctx.prof_get(0)[2]
        gpu_ctx.set_roi(True)
        gpu_ctx.prof_reset()
        win = dm.denoise_3d(t, patch, pad).cpu().numpy()
        fl_win = gpu_ctx.prof_get(0)[2]
    finally:
        gpu_ctx.set_roi(True)
        gpu_ctx.prof_enable(False)
    assert np.isfinite(full).all()
    assert np.array_equal(full, win)
    assert dm.split_stats()[2] == 0
    print(f'{case}: {fl_full / 1e12:.2f} TFLOP in full, {fl_win / 1e12:.2f} with windows')
    if case == 'nf48_odd_levels':
        assert fl_win == fl_full
    else:
        assert fl_win < (0.4 if case == 'nf48_96_48' else 0.8) * fl_full


@pytest.mark.parametrize('case', ['golden_nf8', 'nf48_ragged', 'nf48_odd_levels', 'nf48_96_48'])
def test_tile_windows_3d_on_the_fp32_kernels_are_bit_identical(gpu_ctx, case):
    """round 5: the fp32 kernels of a 3-D program take boxes too (ConvArgs::wz0 / wz1; conv_mfma DIMS = 3, the per-parity fp32
    launches, the 1-output-channel tiled kernel), so exact mode (tpz_ctx_set_exact) and an overflow re-run of a tiled tomogram
    compute what the kept voxels depend on instead of every (patch + 2 * padding)^3 tile in full.  Identical bits with the
    switch off, and the launches do execute less; also a program whose decoder is not an exact 2x upsampling at some level
    (fused-loader kernels: windowed like any other fp32 launch)."""
    from topaz_amd.denoise import Denoise3D
    from topaz_amd.denoising.models import DenoiseNet
    if case == 'golden_nf8':
        z = load_golden('denoise3d_unet3d_nf8')
        d = Denoise3D(DenoiseNet('unet-3d', golden_sd(z)))
        vol, patch, pad = z['tomo'], 32, 16
    else:
        d = Denoise3D(DenoiseNet('unet-3d', oden.synthetic_unet_sd(13, nf=48, base_width=7, top_width=3, dims=3)))
        if case == 'nf48_ragged':
            vol, patch, pad = (np.random.RandomState(31).randn(50, 77, 90) * 2 + 1).astype(np.float32), 32, 16
        elif case == 'nf48_odd_levels':
            vol, patch, pad = (np.random.RandomState(33).randn(45, 50, 85) * 2 + 1).astype(np.float32), 40, 20
        else:
            vol, patch, pad = np.random.RandomState(32).randn(100, 200, 120).astype(np.float32), 96, 48
    dm = d.model.device_model
    t = torch.from_numpy(vol).cuda()
    split = dm.denoise_3d(t, patch, pad).cpu().numpy()           # the default path, for reference
    try:
        gpu_ctx.set_exact(True)
        gpu_ctx.set_roi(False)
        gpu_ctx.prof_enable(1); gpu_ctx.prof_reset()
        full = dm.denoise_3d(t, patch, pad).cpu().numpy()
        fl_full = gpu_ctx.prof_get(0)[2] + gpu_ctx.prof_get(1)[2]
        gpu_ctx.set_roi(True)
        gpu_ctx.prof_reset()
        win = dm.denoise_3d(t, patch, pad).cpu().numpy()
        fl_win = gpu_ctx.prof_get(0)[2] + gpu_ctx.prof_get(1)[2]
    finally:
        gpu_ctx.set_roi(True)
        gpu_ctx.set_exact(False)
        gpu_ctx.prof_enable(False)
    assert np.isfinite(full).all()
    assert np.array_equal(full, win)
    assert _err(win, split) <= ATOL
    if case == 'golden_nf8':
        assert _err(win, load_golden('denoise3d_unet3d_nf8')['p32_16']) <= ATOL
    print(f'{case} (fp32 kernels): {fl_full / 1e12:.2f} TFLOP in full, {fl_win / 1e12:.2f} with windows')
    assert fl_win < (0.4 if case == 'nf48_96_48' else 0.85) * fl_full


def test_batched_tiles_with_mixed_source_chunks(gpu_ctx, monkeypatch):
    """A two-source plane-stacked launch whose chunks mix both tensors (conv_split MODE 3: widths of a user-trained 3-D U-Net
    whose first source does not fill whole chunks; forced here by the debug switch TPZ_NO_SRCMAJOR -- honoured under TPZ_DEBUG=1 only, read when the model is loaded) has no
    batched instantiation: rec_flush must issue those launches one by one instead of failing the whole batched pass with
    'conv_split launch failed: invalid value' (round-4 advisor finding).  Same bits as the unbatched lanes path and as the
    source-major model."""
    from topaz_amd.denoise import Denoise3D
    from topaz_amd.denoising.models import DenoiseNet
    z = load_golden('denoise3d_unet3d_nf8')
    tomo = torch.from_numpy(z['tomo']).cuda()
    ref = Denoise3D(DenoiseNet('unet-3d', golden_sd(z))).model.device_model.denoise_3d(tomo, 32, 16).cpu().numpy()
    monkeypatch.setenv('TPZ_DEBUG', '1')
    monkeypatch.setenv('TPZ_NO_SRCMAJOR', '1')
    d = Denoise3D(DenoiseNet('unet-3d', golden_sd(z)))
    monkeypatch.delenv('TPZ_NO_SRCMAJOR')
    monkeypatch.delenv('TPZ_DEBUG')
    try:
        gpu_ctx.set_batch(8)
        batched = d.model.device_model.denoise_3d(tomo, 32, 16).cpu().numpy()
        gpu_ctx.set_batch(0)
        single = d.model.device_model.denoise_3d(tomo, 32, 16).cpu().numpy()
    finally:
        gpu_ctx.set_batch(8)
    assert np.isfinite(batched).all()
    assert np.array_equal(batched, single)
    assert _err(batched, ref) <= 2e-6          # (another cell order of the K loop: another summation order)
    assert _err(batched, z['p32_16']) <= ATOL


@pytest.mark.parametrize('case', ['bench_net', 'pretrained', 'small', 'fcnn', 'ragged', 'unet3d'])
def test_batched_patches_are_bit_identical(gpu_ctx, case):
    """the same layer of up to 8 patches / tiles in ONE launch (conv_split_multi_kernel, rt_core.hip rec_flush) against one
    patch at a time: identical bits -- patches of different sizes (corner / edge / interior), patch counts that do not fill
    the last batch, batch sizes 2, 3 and 8, windows on, the batches alternating on the two patch lanes (the default) and
    all on the ctx stream; the 3-D tiles through the plane-stacked modes"""
    from topaz_amd.denoise import Denoise, Denoise3D
    from topaz_amd.denoising.models import DenoiseNet
    if case == 'unet3d':
        z = load_golden('denoise3d_unet3d_nf8')
        d = Denoise3D(DenoiseNet('unet-3d', golden_sd(z)))
        tomo = torch.from_numpy(z['tomo']).cuda()
        run = lambda: d.model.device_model.denoise_3d(tomo, 32, 16).cpu().numpy()
    else:
        if case == 'bench_net':
            d = Denoise(DenoiseNet('unet', oden.synthetic_unet_sd(11, nf=48, base_width=11, top_width=5)))
            shape, patch, pad = (1500, 1330), 400, 300
        elif case == 'pretrained':
            d = Denoise('unet-v0.2.1')
            shape, patch, pad = (1111, 1201), 256, 280
        elif case == 'small':
            d = Denoise('unet-small')
            shape, patch, pad = (700, 900), 200, 120
        elif case == 'fcnn':
            d = Denoise('fcnn')
            shape, patch, pad = (500, 640), 128, 64
        else:
            d = Denoise('unet-v0.2.1')
            shape, patch, pad = (333, 1001), 97, 51
        x = (np.random.RandomState(78).randn(*shape) * 3 + 1).astype(np.float32)
        run = lambda: d.denoise(x, patch, pad)
    try:
        gpu_ctx.set_batch(0)
        n0 = gpu_ctx.launches()
        one = run()
        n_one = gpu_ctx.launches() - n0
        outs = {}
        for b in (2, 3, 8):
            gpu_ctx.set_batch(b)
            n0 = gpu_ctx.launches()
            outs[b] = run()
            if b == 8:
                n_batched = gpu_ctx.launches() - n0
        gpu_ctx.set_lanes(False)
        outs['8, lanes off'] = run()
    finally:
        gpu_ctx.set_batch(8)
        gpu_ctx.set_lanes(True)
    assert np.isfinite(one).all()
    for b, y in outs.items():
        assert np.array_equal(one, y), (case, b)
    print(f'{case}: {n_one} launches one patch at a time, {n_batched} batched by 8')
    assert n_batched < n_one


def test_fcnn_runs_wholly_on_the_2xf16_path(gpu_ctx):
    """the pretrained fully convolutional denoiser (DenoiseNet2(64, width 11), denoising/models.py:52-66,597-598): 11x11 stem
    and last conv as column kernels, the 11x11 64->64 body on its own 2xf16 tile -- no layer left on an fp32 kernel; output
    within 1e-4 of the reference's golden and of the exact-fp32 kernels"""
    import warnings
    from topaz_amd.denoise import Denoise
    with warnings.catch_warnings():
        warnings.simplefilter('error')                    # (the mixed-program warning must not fire)
        d = Denoise('fcnn')
    n_conv, n_split, off = d.model.device_model.split_layers()
    assert n_conv == 3 and n_split == 3, off
    z = load_golden('denoise2d_pretrained')
    y = d.denoise(z['x'], 64, 24)
    assert _err(y, z['fcnn:p64_24']) <= ATOL
    x = (np.random.RandomState(3).randn(700, 810) * 2 + 0.5).astype(np.float32)
    a = d.denoise(x, 256, 100)
    before = d.model.device_model.split_stats()
    gpu_ctx.set_exact(True)
    try:
        b = d.denoise(x, 256, 100)
    finally:
        gpu_ctx.set_exact(False)
    assert _err(a, b) <= ATOL and before[2] == 0


def test_unet_nf32_weights_resident_layers_with_windows_and_lanes(gpu_ctx):
    """A user-trained U-Net of 32 filters: its encoder convs (3x3, 32 -> 32) are the shape the weights-resident kernel takes
    (csrc/conv_rw.h) -- here inside a PATCHED denoise, i.e. with launch windows (each patch computes what its kept centre depends
    on) and, unbatched, on the two patch lanes.  A batched pass keeps those layers on the general 2xf16 tile (its launches are
    merged across patches).  Unbatched with windows == unbatched without (bit-identical); unbatched vs batched to rounding
    (another summation order in those layers); both within 1e-4 of the oracle."""
    from topaz_amd.denoise import Denoise
    from topaz_amd.denoising.models import DenoiseNet
    sd = oden.synthetic_unet_sd(17, nf=32, base_width=11, top_width=5)
    d = Denoise(DenoiseNet('unet', sd))
    x = (np.random.RandomState(5).randn(700, 820) * 1.5 + 0.3).astype(np.float32)
    ref = oden.denoise('unet', sd, x, 256, 96)
    try:
        gpu_ctx.set_batch(0)
        gpu_ctx.prof_enable(1); gpu_ctx.prof_reset()
        a = d.denoise(x, 256, 96)
        names = [k[0] for k in gpu_ctx.prof_kernels()]
        gpu_ctx.prof_enable(False)
        gpu_ctx.set_roi(False)
        a_full = d.denoise(x, 256, 96)
        gpu_ctx.set_roi(True)
        gpu_ctx.set_rw(False)
        a_gen = d.denoise(x, 256, 96)
        gpu_ctx.set_rw(True)
        gpu_ctx.set_batch(8)
        b = d.denoise(x, 256, 96)
    finally:
        gpu_ctx.set_batch(8)
        gpu_ctx.set_roi(True)
        gpu_ctx.set_rw(True)
        gpu_ctx.prof_enable(False)
    assert any('conv_split_rw_kernel' in n for n in names), names
    assert np.array_equal(a, a_full)
    assert _err(a, ref) <= ATOL and _err(b, ref) <= ATOL and _err(a_gen, ref) <= ATOL
    assert _err(a, b) <= 2e-5 and np.array_equal(a_gen, b)


@pytest.mark.parametrize('nf', [20, 12, 40])
def test_unet_of_any_width_runs_wholly_on_the_2xf16_path(gpu_ctx, nf):
    """UDenoiseNet (skip connections at every level: decoder sources of 2 nf + nf channels) with widths that are not multiples
    of 16: loaded zero-padded (rt_load.hip widen_program) -- every layer on the f16 matrix cores, whole image and patched,
    within 1e-4 of the oracle, and equal to the exact-fp32 kernels within the same bound."""
    import warnings
    from topaz_amd.denoise import Denoise
    from topaz_amd.denoising.models import DenoiseNet
    sd = oden.synthetic_unet_sd(40 + nf, nf=nf, base_width=11, top_width=5)
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        d = Denoise(DenoiseNet('unet', sd))
    n_conv, n_split, off = d.model.device_model.split_layers()
    assert n_split == n_conv, off
    x = (np.random.RandomState(nf).randn(300, 420) * 1.2 - 0.4).astype(np.float32)
    ref_whole, ref_p = oden.denoise('unet', sd, x, -1), oden.denoise('unet', sd, x, 128, 64)
    a, b = d.denoise(x, -1), d.denoise(x, 128, 64)
    assert _err(a, ref_whole) <= ATOL and _err(b, ref_p) <= ATOL
    gpu_ctx.set_exact(True)
    try:
        a32 = d.denoise(x, -1)
    finally:
        gpu_ctx.set_exact(False)
    assert _err(a, a32) <= ATOL and dm_reruns(d) == 0


def dm_reruns(d):
    return d.model.device_model.split_stats()[2]
