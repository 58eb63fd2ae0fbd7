"""The oracle (oracle/) against every golden vector generated from the reference itself
(oracle/make_golden.py).  CPU only.  Tolerances: the oracle calls the same torch-CPU primitives as
the reference, so scores / pixels agree to float32 round-off (<= 2e-5 absolute on logits spanning
+-20 and pixels ~10); NMS tables are compared exactly."""
import os

import numpy as np
import pytest

from conftest import golden_sd, load_golden
from oracle import denoising, nms, scoring

PRETRAINED = os.path.join(os.path.dirname(__file__), '..', 'topaz_amd', 'pretrained')


def _load_sav(rel):
    import torch
    sd = torch.load(os.path.join(PRETRAINED, rel), map_location='cpu', weights_only=True)
    return {k: v.numpy() for k, v in sd.items()}


@pytest.mark.parametrize('name', ['resnet8_u32', 'resnet16_u32'])
def test_scoring_pretrained(name):
    z = load_golden(f'score_{name}')
    sd = _load_sav(f'detector/{name}.sav')
    arch = str(z['arch'])
    for k in ('0', '1'):
        y = scoring.score(arch, sd, z['x' + k])
        assert y.shape == z['y' + k].shape
        np.testing.assert_allclose(y, z['y' + k], atol=2e-5, rtol=0)


@pytest.mark.parametrize('name', ['resnet8_bn_u16', 'resnet16_u16', 'conv127_bn_u16', 'conv31_u32', 'resnet8_3d_u8',
                                  'resnet8_3d_bn_u8', 'resnet16_3d_u8', 'conv31_3d_bn_u8', 'conv63_3d_bn_u8', 'conv127_3d_bn_u8'])
def test_scoring_seeded(name):
    z = load_golden(f'score_{name}')
    y = scoring.score(str(z['arch']), golden_sd(z), z['x0'])
    np.testing.assert_allclose(y, z['y0'], atol=2e-5, rtol=0)


@pytest.mark.parametrize('name', ['resnet6_u16', 'resnet8_pool_bn_u16', 'resnet16_pool_u8'])
def test_scoring_pooled_resnets(name):
    """ResNet6 and the --pooling max ResNets: filled MaxPool = dilated stride-1 max (resnet.py:30-36)"""
    z = load_golden(f'score_{name}')
    y = scoring.score(str(z['arch']), golden_sd(z), z['x0'], pooling=True)
    np.testing.assert_allclose(y, z['y0'], atol=2e-5, rtol=0)
    spec = scoring.ARCH_SPECS[str(z['arch'])](True)
    assert scoring.width_of(spec) == int(z['width']) and scoring.fill(spec) == 4
    assert [m['pool_dil'] for m in spec if m['type'] == 'pool'] == [1, 2]


def test_fill_geometry():
    s8 = scoring.resnet8_spec()
    assert scoring.width_of(s8) == 71 and scoring.fill(s8) == 4
    assert [(m.get('conv0_fdil'), m.get('conv1_fdil')) for m in s8 if m['type'] == 'resid'] == [(2, 4), (2, 4), (4, 8)]
    assert s8[0]['conv_dil'] == 1 and s8[-1]['conv_dil'] == 4
    s16 = scoring.resnet16_spec()
    assert scoring.width_of(s16) == 91 and scoring.fill(s16) == 4
    assert [(m['conv0_fdil'], m['conv1_fdil']) for m in s16 if m['type'] == 'resid'] == \
        [(1, 1), (2, 2), (2, 2), (2, 2), (2, 2), (4, 4), (4, 4)]


def _nms_case_names(z):
    return sorted({k.split(':')[0] for k in z.files if ':' in k})


def test_nms_golden_cases():
    z = load_golden('nms_cases')
    names = _nms_case_names(z)
    assert len(names) >= 16
    for name in names:
        x, r, thr = z[name + ':x'], int(z[name + ':r']), float(z[name + ':thr'])
        if x.ndim == 2:
            s, c = nms.nms2d(x, r, thr)
            # the fixture must not depend on the (undefined) tie order of numpy's argsort
            s2, c2 = nms.nms2d_py(x, r, thr, ties='asc_index')
        else:
            scale = float(z[name + ':scale'])
            s, c = nms.nms3d(x, r, scale, thr)
            s2, c2 = nms.nms3d_py(x, r, scale, thr, ties='asc_index')
        assert np.array_equal(c, z[name + ':coords']), name
        assert np.array_equal(s, z[name + ':scores']), name
        # ... the opposite tie order must give the same picks (only rows of equal score may swap places)
        assert np.array_equal(s2, z[name + ':scores']), f'{name}: fixture is tie-order sensitive'
        assert sorted(map(tuple, c2.tolist())) == sorted(map(tuple, z[name + ':coords'].tolist())), \
            f'{name}: fixture is tie-order sensitive'


def test_nms_c_matches_python_restatement():
    rs = np.random.RandomState(5)
    for shape, r in (((50, 37), 2), ((31, 64), 5), ((7, 5), 9)):
        x = rs.randn(*shape).astype(np.float32)
        x[rs.rand(*shape) < 0.1] = 0.25          # plenty of ties
        a, b = nms.nms2d(x, r, -0.2), nms.nms2d_py(x, r, -0.2)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    v = rs.randn(9, 10, 11).astype(np.float32)
    v[rs.rand(*v.shape) < 0.1] = 0.5
    a, b = nms.nms3d(v, 2, 1.25, -0.1), nms.nms3d_py(v, 2, 1.25, -0.1)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


KINDS = {'unet-v0.2.1': ('unet', 'denoise/unet_L2_v0.2.1.sav'), 'unet-small': ('unet-small', 'denoise/unet_small_L1_v0.2.2.sav'),
         'fcnn': ('fcnn', 'denoise/fcnn_L1_v0.2.2.sav'), 'affine': ('affine', 'denoise/affine_L1_v0.2.2.sav')}


@pytest.mark.parametrize('name', list(KINDS))
def test_denoise2d_pretrained(name):
    z = load_golden('denoise2d_pretrained')
    kind, rel = KINDS[name]
    sd = _load_sav(rel)
    x = z['x']
    np.testing.assert_allclose(denoising.denoise(kind, sd, x, -1), z[f'{name}:whole'], atol=2e-5, rtol=0)
    np.testing.assert_allclose(denoising.denoise(kind, sd, x, 64, 24), z[f'{name}:p64_24'], atol=2e-5, rtol=0)


def test_denoise_image_variants():
    z = load_golden('denoise2d_pretrained')
    x = z['x']
    unet = ('unet', _load_sav(KINDS['unet-v0.2.1'][1]))
    small = ('unet-small', _load_sav(KINDS['unet-small'][1]))
    aff = ('affine', _load_sav(KINDS['affine'][1]))
    np.testing.assert_allclose(denoising.denoise_image([unet], x.copy(), patch_size=96, padding=16),
                               z['image:unet-v0.2.1:p96_16'], atol=3e-5, rtol=0)
    np.testing.assert_allclose(denoising.denoise_image([unet], x.copy(), normalize=True),
                               z['image:unet-v0.2.1:norm'], atol=3e-5, rtol=0)
    np.testing.assert_allclose(denoising.denoise_image([small], x.copy(), gaus_sigma=1.2),
                               z['image:unet-small:gaus1.2'], atol=3e-5, rtol=0)
    np.testing.assert_allclose(denoising.denoise_image([aff], x.copy(), cutoff=1.5),
                               z['image:affine:cutoff'], atol=3e-5, rtol=0)


def test_denoise2d_seeded_v022_arch():
    z = load_golden('denoise2d_unet_b11t5_nf16')
    sd = golden_sd(z)
    np.testing.assert_allclose(denoising.denoise('unet', sd, z['x'], -1), z['whole'], atol=2e-5, rtol=0)
    np.testing.assert_allclose(denoising.denoise('unet', sd, z['x'], 48, 20), z['p48_20'], atol=2e-5, rtol=0)


@pytest.mark.parametrize('tag,kind', [('unet2_nf12', 'unet2'), ('unet3', 'unet3')])
def test_denoise2d_user_trainable_archs(tag, kind):
    """UDenoiseNet2 (--arch unet2: no skip into dec2 / dec1) and UDenoiseNet3 (--arch unet3: x - dec1(h)),
    denoising/models.py:247-449: the oracle against the reference's own modules, weights read from the full-module pickle
    through topaz_amd's allow-listing unpickler (no reference import)"""
    import os
    import torch
    from collections import OrderedDict
    from conftest import GOLDEN
    from topaz_amd.model.unpickle import _PickleModule, _walk
    z = load_golden(f'denoise2d_{tag}')
    obj = torch.load(os.path.join(GOLDEN, f'user_model_{tag}.sav'), map_location='cpu', weights_only=False, pickle_module=_PickleModule)
    sd = OrderedDict()
    _walk(obj, '', sd)
    assert np.abs(denoising.denoise(kind, sd, z['x'], -1) - z['whole']).max() <= 2e-5
    assert np.abs(denoising.denoise(kind, sd, z['x'], 96, 24) - z['p96_24']).max() <= 2e-5


def test_denoise3d_seeded():
    z = load_golden('denoise3d_unet3d_nf8')
    sd = golden_sd(z)
    np.testing.assert_allclose(denoising.denoise3d(sd, z['tomo'], 32, 16), z['p32_16'], atol=2e-5, rtol=0)
    np.testing.assert_allclose(denoising.denoise3d(sd, z['small'], -1, 0), z['small_whole'], atol=2e-5, rtol=0)


def test_downsample_golden():
    z = load_golden('downsample_cases')
    for name in sorted({k.split(':')[0] for k in z.files if ':' in k}):
        y = denoising.downsample(z[name + ':x'], int(z[name + ':factor']))
        assert y.shape == z[name + ':y'].shape and y.dtype == z[name + ':y'].dtype
        np.testing.assert_allclose(y, z[name + ':y'], atol=1e-6, rtol=0)


@pytest.mark.parametrize('name', ['normalize_s1', 'normalize_s4'])
def test_normalize_oracle_vs_reference(name):
    """oracle/stats.py (float64 NumPy) against topaz.stats.normalize run by the reference (float32 torch):
    all twelve mixture fits, the selected model and the normalised image."""
    from oracle import stats as ost
    g = load_golden(name)
    x, sample, seed = g['x'], int(g['sample']), int(g['seed'])
    xs, scale = x, 1.0
    if sample > 1:
        np.random.seed(seed)
        n = int(np.round(x.size / sample))
        scale = x.size / n
        xs = np.random.choice(x.ravel(), size=n, replace=False)
    mus, stds, pis, logps = ost.norm_fit(xs, 900, 1, scale=scale)
    assert np.abs(mus - g['mus']).max() <= 1e-5 * np.abs(g['mus']).max()
    assert np.abs(stds - g['stds']).max() <= 1e-5 * g['stds'].max()
    assert np.abs(pis - g['pis']).max() <= 3e-4
    assert (np.abs(logps - g['logps']) / np.abs(g['logps'])).max() <= 1e-5
    i = int(np.argmax(logps))
    assert i == int(np.argmax(g['logps']))
    y = ((x - mus[i]) / stds[i]).astype(np.float32)
    assert np.abs(y - g['y']).max() <= 2e-6
