"""Generate the golden vectors under tests/golden/ by running the REFERENCE itself.

Test infrastructure; runs only in the build container, where /root/reference (tbepler/topaz
v0.3.18) is mounted.  Recipe:

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py

The reference is imported with one stub (h5py, imported at module top by
topaz/denoising/datasets.py:8 but unused on this path).  Nothing of the reference is copied:
the fixtures hold seeded inputs, the reference's outputs, and -- for architectures whose
pretrained blobs are missing -- the seeded weights that were loaded into the reference modules.
"""
from __future__ import annotations

import os
import sys
import types

sys.dont_write_bytecode = True
REF = os.environ.get('TOPAZ_REFERENCE', '/root/reference')
sys.path.insert(0, REF)
_h5 = types.ModuleType('h5py')
_h5.File = object
sys.modules['h5py'] = _h5

import numpy as np  # noqa: E402
import torch  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden')
META = dict(reference='tbepler/topaz 0.3.18', torch=torch.__version__, numpy=np.__version__)


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, meta=np.asarray(repr(META)), **arrays)
    print(f'{name}: {os.path.getsize(path) / 1024:.0f} KiB')


def sd_arrays(model, prefix='sd:'):
    return {prefix + k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}


def randomise_bn(model, seed):
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm3d)):
            m.weight.data = 1.0 + 0.1 * torch.randn(m.weight.shape, generator=g)
            m.bias.data = 0.1 * torch.randn(m.bias.shape, generator=g)
            m.running_mean.data = 0.1 * torch.randn(m.running_mean.shape, generator=g)
            m.running_var.data = 1.0 + 0.2 * torch.rand(m.running_var.shape, generator=g)


def image(seed, h, w):
    return np.random.RandomState(seed).randn(h, w).astype(np.float32)


# ------------------------------------------------------------------------------------------------
def scoring():
    from topaz.model.factory import load_model
    from topaz.model.classifier import LinearClassifier
    from topaz.model.features.resnet import ResNet8, ResNet16
    from topaz.model.features.basic import BasicConv

    def run(model, x):
        model.eval()
        if not getattr(model, '_golden_filled', False):
            model.fill()                            # once, as extract.py:230 does (a second fill() compounds)
            model._golden_filled = True
        with torch.no_grad():                       # extract.py:249 path (run under no_grad)
            return model(torch.from_numpy(x)[None, None])[0, 0].numpy()

    for name in ('resnet8_u32', 'resnet16_u32'):
        m = load_model(name)
        xs = {'x0': image(1, 96, 96), 'x1': image(2, 160, 200)}
        ys = {k.replace('x', 'y'): run(m, v) for k, v in xs.items()}
        save(f'score_{name}', arch=np.asarray(name.split('_')[0]), **xs, **ys)

    # seeded nets (random init of the reference's own constructors)
    torch.manual_seed(7)
    m = LinearClassifier(ResNet8(units=16, bn=True))
    randomise_bn(m, 8)
    torch.save(m, os.path.join(OUT, 'user_model_resnet8_bn_u16.sav'))   # what `topaz train` writes (training.py:601)
    x = image(3, 120, 136)
    save('score_resnet8_bn_u16', arch=np.asarray('resnet8'), x0=x, y0=run(m, x), **sd_arrays(m))

    torch.manual_seed(9)
    m = LinearClassifier(ResNet16(units=16, bn=False))
    x = image(4, 100, 112)
    save('score_resnet16_u16', arch=np.asarray('resnet16'), x0=x, y0=run(m, x), **sd_arrays(m))

    torch.manual_seed(11)
    m = LinearClassifier(BasicConv([7, 5, 5, 5, 5], 16, bn=True))      # factory.py:15-17 conv127
    randomise_bn(m, 12)
    for p in m.modules():
        if isinstance(p, torch.nn.PReLU):
            p.weight.data.uniform_(0.1, 0.4)
    torch.save(m, os.path.join(OUT, 'user_model_conv127_bn_u16.sav'))
    x = image(5, 140, 150)
    save('score_conv127_bn_u16', arch=np.asarray('conv127'), x0=x, y0=run(m, x), **sd_arrays(m))

    torch.manual_seed(13)
    m = LinearClassifier(BasicConv([7, 5, 5], 32, bn=False))           # conv31, bias instead of BN
    x = image(6, 64, 72)
    save('score_conv31_u32', arch=np.asarray('conv31'), x0=x, y0=run(m, x), **sd_arrays(m))

    # patched scoring (model/utils.py:110-193), float64 result
    from topaz.model.utils import predict_in_patches
    m = load_model('resnet8_u32')
    m.eval()
    m.fill()
    x = image(14, 200, 260)
    y = predict_in_patches(m, torch.from_numpy(x)[None, None], 96 + 2 * (m.width // 2))
    save('score_patched_resnet8_u32', x0=x, y0=y[0, 0], patch=np.asarray(96))


def nms():
    from topaz.algorithms import non_maximum_suppression, non_maximum_suppression_3d
    from topaz.model.factory import load_model
    cases = {}

    def add2(name, x, r, thr):
        s, c = non_maximum_suppression(x, r, threshold=thr)
        cases[name + ':x'] = x
        cases[name + ':r'] = np.asarray(r)
        cases[name + ':thr'] = np.asarray(thr, dtype=np.float64)
        cases[name + ':scores'] = s
        cases[name + ':coords'] = c

    def add3(name, x, r, scale, thr):
        s, c = non_maximum_suppression_3d(x, r, scale=scale, threshold=thr)
        cases[name + ':x'] = x
        cases[name + ':r'] = np.asarray(r)
        cases[name + ':scale'] = np.asarray(scale, dtype=np.float64)
        cases[name + ':thr'] = np.asarray(thr, dtype=np.float64)
        cases[name + ':scores'] = s
        cases[name + ':coords'] = c

    rs = np.random.RandomState(21)
    add2('rand_r3', rs.randn(64, 80).astype(np.float32), 3, -0.5)
    add2('rand_r1', rs.randn(40, 33).astype(np.float32), 1, 0.0)
    add2('rand_r0', rs.randn(9, 11).astype(np.float32), 0, 0.5)
    add2('rand_noinf', rs.randn(24, 24).astype(np.float32), 4, -np.inf)
    # right-edge wrap: the peak at (y=10, x=W-1) suppresses (y=11, x=0) with r=3 ...
    x = np.full((20, 30), -10, dtype=np.float32)
    x[10, 29] = 5.0
    x[11, 0] = 4.0
    x[12, 0] = 3.5
    x[14, 0] = 3.0
    add2('wrap_edge', x, 3, -6.0)
    # ... but a peak at x = W-6 does not
    x = np.full((20, 30), -10, dtype=np.float32)
    x[10, 24] = 5.0
    x[11, 0] = 4.0
    add2('wrap_far', x, 3, -6.0)
    # strict threshold: -5.9 kept at t=-6, -6.0 dropped
    x = np.full((8, 8), -7, dtype=np.float32)
    x[2, 2] = -5.9
    x[5, 5] = -6.0
    add2('thr_strict', x, 1, -6.0)
    # bottom/right corner and low-side clipping
    x = rs.randn(16, 12).astype(np.float32)
    x[15, 11] = 9
    x[0, 0] = 8
    x[0, 11] = 7
    x[15, 0] = 6
    add2('corners', x, 5, -1.0)
    # tiny image, radius larger than the image
    add2('tiny', rs.randn(3, 4).astype(np.float32), 6, -5.0)
    # real logit map, several radii
    m = load_model('resnet8_u32')
    m.eval()
    m.fill()
    with torch.no_grad():
        logit = m(torch.from_numpy(image(2, 160, 200))[None, None])[0, 0].numpy()
    for r in (1, 3, 8, 14):
        add2(f'logits_r{r}', logit, r, -6.0)
    # 3-D
    add3('vol_r2', rs.randn(12, 20, 16).astype(np.float32), 2, 1.0, 0.0)
    add3('vol_r2_s15', rs.randn(10, 12, 14).astype(np.float32), 2, 1.5, -0.3)
    v = np.full((6, 8, 10), -10, dtype=np.float32)
    v[2, 2, 9] = 5.0
    v[2, 3, 0] = 4.0          # wraps: flat-index delta suppresses across the row
    v[4, 7, 9] = 3.0
    v[5, 0, 0] = 2.5
    add3('vol_wrap', v, 1, 1.0, -6.0)
    save('nms_cases', **cases)


def denoise2d():
    from topaz.denoise import Denoise, denoise_image
    from topaz.filters import GaussianDenoise
    out = {}
    x = image(31, 150, 140) * 3.0 + 10.0
    out['x'] = x
    for name in ('unet-v0.2.1', 'unet-small', 'fcnn', 'affine'):
        d = Denoise(name)
        out[f'{name}:whole'] = d.denoise(x, patch_size=-1)
        out[f'{name}:p64_24'] = d.denoise(x, patch_size=64, padding=24)
    d = Denoise('unet-v0.2.1')
    out['image:unet-v0.2.1:p96_16'] = denoise_image(x.copy(), [d], patch_size=96, padding=16)
    out['image:unet-v0.2.1:norm'] = denoise_image(x.copy(), [d], patch_size=-1, padding=0, normalize=True)
    g = GaussianDenoise(1.2)
    out['image:unet-small:gaus1.2'] = denoise_image(x.copy(), [Denoise('unet-small')], gaus=g, patch_size=-1)
    out['gaus1.2:apply'] = g.apply(x)
    out['image:affine:cutoff'] = denoise_image(x.copy(), [Denoise('affine')], cutoff=1.5, patch_size=-1)
    # seeded v0.2.2 architecture (blob missing): UDenoiseNet(base 11, top 5)
    from topaz.denoising.models import UDenoiseNet
    torch.manual_seed(41)
    net = UDenoiseNet(nf=16, base_width=11, top_width=5)
    dn = Denoise.__new__(Denoise)
    dn.model, dn.device, dn.dims, dn.use_cuda = net.eval(), torch.device('cpu'), 2, False
    torch.save(net, os.path.join(OUT, 'user_model_unet_b11t5_nf16.sav'))   # denoising/models.py:628-633 save_model
    x2 = image(32, 110, 121)
    save('denoise2d_unet_b11t5_nf16', x=x2, whole=dn.denoise(x2, patch_size=-1), p48_20=dn.denoise(x2, 48, 20),
         **sd_arrays(net))
    save('denoise2d_pretrained', **out)


def denoise3d():
    from topaz.denoise import Denoise3D
    from topaz.denoising.models import UDenoiseNet3D
    torch.manual_seed(51)
    net = UDenoiseNet3D(nf=8, base_width=7)
    dn = Denoise3D.__new__(Denoise3D)
    dn.model, dn.device, dn.dims, dn.use_cuda = net.eval(), torch.device('cpu'), 3, False
    tomo = (np.random.RandomState(52).randn(40, 50, 60) * 2.0 + 1.0).astype(np.float32)
    y = dn.denoise(tomo, patch_size=32, padding=16, verbose=False)
    small = tomo[:33, :34, :36].copy()
    y_whole = dn.denoise(small, patch_size=-1, verbose=False)
    save('denoise3d_unet3d_nf8', tomo=tomo, p32_16=y, small=small, small_whole=y_whole, **sd_arrays(net))


def cli():
    """files either side of the path: MRC in, pick tables / denoised MRC out (reference CLI + file drivers)"""
    import io
    import contextlib
    import shutil
    import tempfile
    from topaz import mrc
    import topaz.commands.extract as cmd_extract     # topaz.main imports train -> torchvision (absent)

    def topaz_extract(argv):
        cmd_extract.main(cmd_extract.add_arguments().parse_args(argv))
    from topaz.denoise import Denoise, Denoise3D, denoise_stream, denoise_tomogram_stream
    from topaz.denoising.models import UDenoiseNet3D
    tmp = tempfile.mkdtemp()
    cli_dir = os.path.join(OUT, 'cli')
    os.makedirs(cli_dir, exist_ok=True)
    try:
        for name, seed in (('mic_a', 61), ('mic_b', 62)):
            x = image(seed, 160, 200)
            with open(os.path.join(tmp, name + '.mrc'), 'wb') as f:
                mrc.write(f, x[np.newaxis], ax=1.5, ay=1.5, az=1.0)
            shutil.copy(os.path.join(tmp, name + '.mrc'), os.path.join(cli_dir, name + '.mrc'))
        mics = [os.path.join(tmp, 'mic_a.mrc'), os.path.join(tmp, 'mic_b.mrc')]
        # topaz extract (single TSV, coord format) and per-micrograph star files
        topaz_extract(['-m', 'resnet8_u32', '-r', '8', '-d', '-1', '-o', os.path.join(tmp, 'picks.txt')] + mics)
        shutil.copy(os.path.join(tmp, 'picks.txt'), os.path.join(cli_dir, 'extract_picks.txt'))
        os.makedirs(os.path.join(tmp, 'out', 'COORDS'), exist_ok=True)   # the reference never creates it
        topaz_extract(['-m', 'resnet8_u32', '-r', '8', '-t', '-3', '-x', '2', '-d', '-1', '--per-micrograph',
                       '--format', 'star', '-o', os.path.join(tmp, 'out', 'x'), mics[0]])
        shutil.copy(os.path.join(tmp, 'out', 'COORDS', 'mic_a.star'), os.path.join(cli_dir, 'extract_mic_a.star'))
        # denoise_stream with the v0.2.1 U-Net (the CLI itself can only load the missing `unet` blob)
        denoise_stream(mics[:1], os.path.join(tmp, 'den'), format='mrc', suffix='', models=[Denoise('unet-v0.2.1')],
                       deconvolve=False, patch_size=96, padding=24, normalize=False)
        shutil.copy(os.path.join(tmp, 'den', 'mic_a.mrc'), os.path.join(cli_dir, 'denoise_mic_a.mrc'))
        # a small tomogram through denoise_tomogram_stream with a seeded 3-D net saved the way the CLI loads it
        torch.manual_seed(71)
        net = UDenoiseNet3D(nf=8, base_width=7)
        torch.save(net.state_dict(), os.path.join(cli_dir, 'unet3d_nf8_state.sav'))
        dn = Denoise3D.__new__(Denoise3D)
        dn.model, dn.device, dn.dims, dn.use_cuda = net.eval(), torch.device('cpu'), 3, False
        tomo = (np.random.RandomState(72).randn(36, 40, 44)).astype(np.float32)
        with open(os.path.join(tmp, 'tomo.mrc'), 'wb') as f:
            mrc.write(f, tomo)
        shutil.copy(os.path.join(tmp, 'tomo.mrc'), os.path.join(cli_dir, 'tomo.mrc'))
        denoise_tomogram_stream([os.path.join(tmp, 'tomo.mrc')], dn, os.path.join(tmp, 'den3'), gaus=0, patch_size=32,
                                padding=16, verbose=False)
        shutil.copy(os.path.join(tmp, 'den3', 'tomo.mrc'), os.path.join(cli_dir, 'denoise3d_tomo.mrc'))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    for fn in sorted(os.listdir(cli_dir)):
        print('cli/' + fn, os.path.getsize(os.path.join(cli_dir, fn)))


def scoring3d():
    """3-D scoring (`--dims 3`): LinearClassifier(ResNet8(dims=3)) seeded, filled, on small tomograms, and
    classify_patches (classifier.py:69-102) over PatchDataset tiles; plus non_maximum_suppression_3d of the map."""
    from topaz.algorithms import non_maximum_suppression_3d
    from topaz.model.classifier import LinearClassifier, classify_patches
    from topaz.model.features.resnet import ResNet8, ResNet16
    rs = np.random.RandomState(91)
    for name, ctor, units, bn, shape in (('resnet8_3d_u8', ResNet8, 8, False, (20, 26, 30)),
                                         ('resnet8_3d_bn_u8', ResNet8, 8, True, (18, 20, 24)),
                                         ('resnet16_3d_u8', ResNet16, 8, False, (14, 18, 20))):
        torch.manual_seed(92 + len(name))
        m = LinearClassifier(ctor(dims=3, units=units, bn=bn), dims=3)
        if bn:
            randomise_bn(m, 93)
        m.eval()
        m.fill()
        x = rs.randn(*shape).astype(np.float32)
        with torch.no_grad():
            y = m(torch.from_numpy(x)[None, None])[0, 0].numpy()
        arrays = dict(arch=np.asarray(name.split('_')[0]), x0=x, y0=y, **sd_arrays(m))
        if name == 'resnet8_3d_u8':
            s, c = non_maximum_suppression_3d(y, 3, threshold=float(np.quantile(y, 0.5)))
            arrays.update(nms_r=np.asarray(3), nms_thr=np.asarray(np.float32(np.quantile(y, 0.5))), nms_scores=s, nms_coords=c)
            with torch.no_grad():
                yp = classify_patches(m, torch.from_numpy(x)[None], patch_size=16, padding=35, batch_size=2, verbose=False)
            arrays.update(patched=yp[0].numpy(), patch_size=np.asarray(16), padding=np.asarray(35))
            torch.save(m, os.path.join(OUT, 'user_model_resnet8_3d_u8.sav'))
        save('score_' + name, **arrays)


def scoring3d_basic():
    """3-D BasicConv stacks (basic.py:12-111 with dims = 3; `topaz train -m conv31 --dims 3` builds them through
    factory.conv31(units, dims=3, ...)): LinearClassifier(BasicConv([7, 5, 5] / [7, 5, 5, 5], dims=3)), BN + PReLU, seeded,
    filled, on small tomograms."""
    from topaz.model.classifier import LinearClassifier
    from topaz.model.features.basic import BasicConv
    rs = np.random.RandomState(95)
    for name, sizes, units, shape in (('conv31_3d_bn_u8', [7, 5, 5], 8, (14, 20, 26)), ('conv63_3d_bn_u8', [7, 5, 5, 5], 8, (10, 12, 40))):
        torch.manual_seed(96 + len(sizes))
        m = LinearClassifier(BasicConv(sizes, units, dims=3), dims=3)
        randomise_bn(m, 97)
        g = torch.Generator().manual_seed(98)
        for mod in m.modules():
            if isinstance(mod, torch.nn.PReLU):
                mod.weight.data = 0.05 + 0.4 * torch.rand(mod.weight.shape, generator=g)
        m.eval()
        m.fill()
        x = rs.randn(*shape).astype(np.float32)
        with torch.no_grad():
            y = m(torch.from_numpy(x)[None, None])[0, 0].numpy()
        save('score_' + name, arch=np.asarray(name.split('_')[0]), x0=x, y0=y, **sd_arrays(m))
        if name.startswith('conv31'):
            torch.save(m, os.path.join(OUT, 'user_model_conv31_3d_bn_u8.sav'))


def scoring3d_conv127():
    """The 3-D conv127 (basic.py:16-30 with dims = 3, factory.py:15-17): LinearClassifier(BasicConv([7, 5, 5, 5, 5], dims=3)), BN +
    PReLU, seeded, filled (dilations 1, 2, 4, 8, 16: receptive field 127^3), on a small tomogram.  A generator of its own so that
    the fixtures of scoring3d_basic stay byte for byte what they were."""
    from topaz.model.classifier import LinearClassifier
    from topaz.model.features.basic import BasicConv
    rs = np.random.RandomState(195)
    torch.manual_seed(196)
    m = LinearClassifier(BasicConv([7, 5, 5, 5, 5], 8, dims=3), dims=3)
    randomise_bn(m, 197)
    g = torch.Generator().manual_seed(198)
    for mod in m.modules():
        if isinstance(mod, torch.nn.PReLU):
            mod.weight.data = 0.05 + 0.4 * torch.rand(mod.weight.shape, generator=g)
    m.eval()
    m.fill()
    x = rs.randn(6, 9, 21).astype(np.float32)
    with torch.no_grad():
        y = m(torch.from_numpy(x)[None, None])[0, 0].numpy()
    save('score_conv127_3d_bn_u8', arch=np.asarray('conv127'), x0=x, y0=y, **sd_arrays(m))


def extras():
    """rows the first round left partial: the radius search / validation of `topaz extract --targets`
    (extract.py:135-204,284-305), `topaz segment` score maps (model/utils.py:71-105), the inverse-Gaussian pre-filter.
    InvGaussianFilter itself cannot be constructed upstream (filters.py:85 calls GaussianDenoise.__init__ without
    sigma -> TypeError), so its golden is assembled from the reference's working parts: gaussian_filter,
    inverse_filter and AffineFilter (filters.py:6-38)."""
    import contextlib
    import io
    import shutil
    import tempfile
    import pandas as pd
    from topaz import mrc
    import topaz.commands.extract as cmd_extract
    from topaz.filters import AffineFilter, gaussian_filter, inverse_filter
    from topaz.model.factory import load_model
    from topaz.model.utils import segment_images
    cli_dir = os.path.join(OUT, 'cli')
    tmp = tempfile.mkdtemp()
    try:
        mics = []
        for name in ('mic_a', 'mic_b'):
            shutil.copy(os.path.join(cli_dir, name + '.mrc'), os.path.join(tmp, name + '.mrc'))
            mics.append(os.path.join(tmp, name + '.mrc'))
        # targets: a subset of the reference's own r = 8 picks, jittered by up to 2 px, plus a few decoys
        picks = pd.read_csv(os.path.join(cli_dir, 'extract_picks.txt'), sep='\t')
        rs = np.random.RandomState(77)
        keep = picks[picks.score > -3.5].copy()
        keep = keep.iloc[rs.permutation(len(keep))[: max(20, len(keep) // 2)]]
        keep['x_coord'] = np.clip(keep.x_coord + rs.randint(-2, 3, len(keep)), 0, 199)
        keep['y_coord'] = np.clip(keep.y_coord + rs.randint(-2, 3, len(keep)), 0, 159)
        decoys = pd.DataFrame({'image_name': ['mic_a'] * 5 + ['mic_b'] * 5, 'x_coord': rs.randint(0, 200, 10),
                               'y_coord': rs.randint(0, 160, 10), 'score': 0.0})
        targets = pd.concat([keep, decoys])[['image_name', 'x_coord', 'y_coord']]
        # the reference keys its score dict by the PATH it was given and looks the targets' image_name up in it
        # (extract.py:284-290): they only meet when the table names the micrographs exactly as the command line does
        targets['image_name'] = targets['image_name'] + '.mrc'

        tpath = os.path.join(cli_dir, 'targets.txt')
        targets.to_csv(tpath, sep='\t', index=False)

        def run(argv):
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                cmd_extract.main(cmd_extract.add_arguments().parse_args(argv))
            return buf.getvalue()

        # (a) radius search: no -r, --targets given -> find_opt_radius over 4..16 step 4, then extraction at the optimum
        # (run from inside the directory so that the paths are the bare file names the targets table uses)
        cwd = os.getcwd()
        os.chdir(tmp)
        try:
            out = run(['-m', 'resnet8_u32', '-d', '-1', '--targets', tpath, '--min-radius', '4', '--max-radius', '16',
                       '--step-radius', '4', '-o', os.path.join(tmp, 'opt_picks.txt'), 'mic_a.mrc', 'mic_b.mrc'])
            with open(os.path.join(cli_dir, 'targets_search_stdout.txt'), 'w') as f:
                f.write(out)
            shutil.copy(os.path.join(tmp, 'opt_picks.txt'), os.path.join(cli_dir, 'targets_search_picks.txt'))
            # (b) validation at a fixed radius with an assignment radius
            out = run(['-m', 'resnet8_u32', '-d', '-1', '-r', '8', '--assignment-radius', '5', '--targets', tpath,
                       '--only-validate', 'mic_a.mrc', 'mic_b.mrc'])
            with open(os.path.join(cli_dir, 'targets_validate_stdout.txt'), 'w') as f:
                f.write(out)
        finally:
            os.chdir(cwd)
        # (c) topaz segment: the score map of mic_a as the float32 TIFF the reference writes
        m = load_model('resnet8_u32')
        m.eval()
        m.fill()
        segment_images(m, mics[:1], os.path.join(tmp, 'seg'), use_cuda=False, verbose=False)
        shutil.copy(os.path.join(tmp, 'seg', 'mic_a.tiff'), os.path.join(cli_dir, 'segment_mic_a.tiff'))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    # (d) inverse Gaussian filter from the reference's working parts
    x = image(95, 90, 120) * 3 + 10
    out = {'x': x}
    for sigma in (0.8, 1.5):
        width = 1 + 2 * int(np.ceil(sigma * 5))
        f = gaussian_filter(sigma, s=width)
        f /= f.sum()
        Fi = inverse_filter(f)
        with torch.no_grad():
            y = AffineFilter(Fi)(torch.from_numpy(x)[None, None])[0, 0].numpy()
        out[f'kernel:{sigma}'] = Fi
        out[f'y:{sigma}'] = y
    save('inv_gaussian', **out)
    for fn in ('targets.txt', 'targets_search_stdout.txt', 'targets_search_picks.txt', 'targets_validate_stdout.txt',
               'segment_mic_a.tiff'):
        print('cli/' + fn, os.path.getsize(os.path.join(cli_dir, fn)))


def pooled():
    """ResNets with MaxPool layers: ResNet6 (always pooled) and `topaz train --pooling max` ResNet8 / ResNet16
    (resnet.py:10-47,254-339), seeded, filled; one full-module pickle for the loader."""
    from topaz.model.classifier import LinearClassifier
    from topaz.model.features.resnet import MaxPool, ResNet6, ResNet8, ResNet16
    rs = np.random.RandomState(101)
    for name, ctor, kw, bn in (('resnet6_u16', ResNet6, {}, False), ('resnet8_pool_bn_u16', ResNet8, dict(pooling=MaxPool), True),
                               ('resnet16_pool_u8', ResNet16, dict(pooling=MaxPool), False)):
        units = int(name.rsplit('u', 1)[1])
        torch.manual_seed(102 + len(name))
        m = LinearClassifier(ctor(units=units, bn=bn, **kw))
        if bn:
            randomise_bn(m, 103)
        if name == 'resnet8_pool_bn_u16':
            torch.save(m, os.path.join(OUT, 'user_model_resnet8_pool_bn_u16.sav'))      # saved unfilled, like training does
        m.eval()
        width = m.width
        m.fill()
        x = rs.randn(120, 150).astype(np.float32)
        with torch.no_grad():
            y = m(torch.from_numpy(x)[None, None])[0, 0].numpy()
        save('score_' + name, arch=np.asarray(name.split('_')[0]), pooling=np.asarray(True), width=np.asarray(width), x0=x, y0=y,
             **sd_arrays(m))


def dropout():
    """`topaz train --dropout p` models: nn.Dropout modules sit between the blocks (resnet.py:296-303, basic.py:57-70) and
    shift the indices of every later module; inference ignores them.  Full-module pickles (saved unfilled, as training
    does) + the reference's own eval-mode scores of the filled nets."""
    from topaz.model.classifier import LinearClassifier
    from topaz.model.features.basic import BasicConv
    from topaz.model.features.resnet import ResNet8
    rs = np.random.RandomState(131)
    out = {}
    for name, make in (('resnet8_drop_bn_u16', lambda: ResNet8(units=16, bn=True, dropout=0.25)),
                       ('conv31_drop_bn_u16', lambda: BasicConv(layers=[7, 5, 5], units=16, bn=True, dropout=0.3))):
        torch.manual_seed(132 + len(name))
        m = LinearClassifier(make())
        randomise_bn(m, 133)
        torch.save(m, os.path.join(OUT, f'user_model_{name}.sav'))
        m.eval()
        m.fill()
        x = rs.randn(96, 130).astype(np.float32)
        with torch.no_grad():
            out[name + ':x'] = x
            out[name + ':y'] = m(torch.from_numpy(x)[None, None])[0, 0].numpy()
    save('score_dropout_models', **out)


def downsample():
    """truncated-DFT downsample (utils/image.py:38-61), the step before the path in `topaz preprocess`"""
    from topaz.utils.image import downsample as ref_downsample
    out = {}
    for name, (h, w), factor in (('even_f2', (200, 260), 2), ('even_f4', (256, 320), 4), ('odd_f3', (201, 263), 3),
                                 ('odd_f8', (333, 250), 8)):
        x = image(81 + h, h, w) * 2 + 1
        out[name + ':x'] = x
        out[name + ':factor'] = np.asarray(factor)
        out[name + ':y'] = ref_downsample(x, factor)
    save('downsample_cases', **out)


def normalize():
    """topaz.stats.normalize (GMM fit, stats.py:37-203) on a bimodal synthetic micrograph: background N(10, 2^2)
    plus 8 % brighter 'particle' pixels; once on every pixel, once on a seeded random quarter of them."""
    from topaz.stats import normalize as ref_normalize
    rs = np.random.RandomState(21)
    x = (rs.randn(180, 200) * 2 + 10).astype(np.float32)
    mask = rs.rand(180, 200) < 0.08
    x[mask] += (4 + rs.randn(int(mask.sum()))).astype(np.float32)
    for sample, seed in ((1, None), (4, 5)):
        if seed is not None:
            np.random.seed(seed)
        y, md = ref_normalize(x.copy(), alpha=900, beta=1, num_iters=100, sample=sample)
        save(f'normalize_s{sample}', x=x, y=y, sample=np.asarray(sample), seed=np.asarray(-1 if seed is None else seed),
             mu=np.asarray(md['mu']), std=np.asarray(md['std']), pi=np.asarray(md['pi']), logp=np.asarray(md['logp']),
             mus=md['mus'], stds=md['stds'], pis=md['pis'], logps=md['logps'])
    y, md = ref_normalize(x.copy(), method='affine')
    save('normalize_affine', x=x, y=y, mu=np.asarray(md['mu']), std=np.asarray(md['std']))


def denoise2d_user_archs():
    """the user-trainable denoiser classes besides UDenoiseNet (`topaz denoise --arch unet2 | unet3`, commands/denoise.py:56;
    denoising/models.py:247-449), seeded: outputs of the reference's own modules + the full-module pickles `train_model` would
    have written (models.py:628-633).  DenoiseNet (--arch fcnet) is not here: its forward pass raises upstream (a 3*nf-channel
    tensor fed to Conv2d(nf, 2*nf), models.py:38-39)."""
    from topaz.denoise import Denoise
    from topaz.denoising.models import UDenoiseNet2, UDenoiseNet3
    x = image(61, 150, 133)
    for tag, ctor, seed in (('unet2_nf12', lambda: UDenoiseNet2(nf=12), 71), ('unet3', lambda: UDenoiseNet3(), 72)):
        torch.manual_seed(seed)
        net = ctor().eval()
        dn = Denoise.__new__(Denoise)
        dn.model, dn.device, dn.dims, dn.use_cuda = net, torch.device('cpu'), 2, False
        torch.save(net, os.path.join(OUT, f'user_model_{tag}.sav'))
        # (the weights travel in the pickle only: tests read them back through topaz_amd's own unpickler)
        save(f'denoise2d_{tag}', x=x, whole=dn.denoise(x, patch_size=-1), p96_24=dn.denoise(x, 96, 24))


if __name__ == '__main__':
    which = sys.argv[1:] or ['scoring', 'nms', 'denoise2d', 'denoise3d', 'cli', 'downsample']
    torch.set_num_threads(8)
    for w in which:
        globals()[w]()
