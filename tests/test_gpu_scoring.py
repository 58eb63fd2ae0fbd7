"""Filled scoring networks on the MI355X against (a) the golden vectors produced by the
reference itself and (b) the CPU oracle on seeded inputs.  Tolerance 1e-4 absolute on logits
(BASELINE.json north_star); end-to-end NMS tables must be identical."""
import numpy as np
import pytest
import torch

from conftest import golden_sd, load_golden
from oracle import nms as onms
from oracle import scoring as oscoring

pytestmark = pytest.mark.gpu
ATOL = 1e-4


def _score(model, x):
    model.eval()
    model.fill()
    model.cuda()
    with torch.no_grad():
        return model(torch.from_numpy(x)[None, None].cuda())[0, 0].cpu().numpy()


@pytest.mark.parametrize('name', ['resnet8_u32', 'resnet16_u32'])
def test_pretrained_vs_reference_golden(gpu_ctx, name):
    from topaz_amd.model.factory import load_model
    z = load_golden(f'score_{name}')
    m = load_model(name)
    assert m.width == {'resnet8_u32': 71, 'resnet16_u32': 91}[name]
    for k in ('0', '1'):
        y = _score(m, z['x' + k])
        assert y.shape == z['y' + k].shape
        assert np.abs(y - z['y' + k]).max() <= ATOL


@pytest.mark.parametrize('name', ['resnet8_bn_u16', 'resnet16_u16', 'conv127_bn_u16', 'conv31_u32'])
def test_seeded_vs_reference_golden(gpu_ctx, name):
    from topaz_amd.model.classifier import LinearClassifier
    z = load_golden(f'score_{name}')
    m = LinearClassifier(str(z['arch']), golden_sd(z))
    y = _score(m, z['x0'])
    assert np.abs(y - z['y0']).max() <= ATOL


def test_prelu_slopes_outside_unit_interval(gpu_ctx):
    """conv31 (BasicConv: PReLU after every conv) with learned slopes the 2xf16 epilogue's max(v, slope * v) cannot express
    (> 1) next to ones it can (negative, 0, 1): a layer with slope > 1 must stay on its fp32 kernel; result vs the oracle"""
    from topaz_amd.model.classifier import LinearClassifier
    z = load_golden('score_conv31_u32')
    sd = dict(golden_sd(z))
    acts = sorted(k for k in sd if k.startswith('features.features.') and np.asarray(sd[k]).size == 1)
    assert len(acts) == 3
    x = z['x0']
    for slopes in ((1.7, -0.4, 0.0), (0.3, 2.5, 1.0), (1.0001, 0.999, 3.0)):
        sd = dict(sd)
        for k, v in zip(acts, slopes):
            sd[k] = np.full_like(np.asarray(sd[k]), v)
        # slopes > 1 amplify the negative lobe layer after layer: the (linear) head is rescaled so that the logits keep the
        # range of a real detector (|logit| <= 20) and the tolerance below is the ABSOLUTE 1e-4 of every other test
        peak = float(np.abs(oscoring.score('conv31', sd, x)).max())
        if peak > 20:
            for k in ('classifier.weight', 'classifier.bias'):
                sd[k] = (np.asarray(sd[k], dtype=np.float64) * (20.0 / peak)).astype(np.float32)
        ref = oscoring.score('conv31', sd, x)
        assert np.abs(ref).max() <= 20.5
        y = _score(LinearClassifier('conv31', sd), x)
        assert np.abs(y - ref).max() <= ATOL, (slopes, float(np.abs(y - ref).max()))


@pytest.mark.parametrize('arch,units', [('resnet8', 64), ('resnet16', 64), ('resnet8', 32)])
def test_seeded_vs_oracle_512(gpu_ctx, arch, units):
    """the CLI-default widths (u64, blobs missing upstream) with seeded weights, 512x384 image"""
    from topaz_amd.model.classifier import LinearClassifier
    sd = oscoring.synthetic_resnet_sd(arch, units, seed=7)
    x = np.random.RandomState(1000).randn(384, 512).astype(np.float32)
    ref = oscoring.score(arch, sd, x)
    y = _score(LinearClassifier(arch, sd), x)
    assert np.abs(y - ref).max() <= ATOL


def test_end_to_end_picks_identical(gpu_ctx):
    """score -> NMS on the device vs oracle score -> oracle NMS: identical coordinate table,
    or every mismatch explained by a score gap below the tolerance (NMS tiers)."""
    from topaz_amd.model.factory import load_model
    from topaz_amd.algorithms import non_maximum_suppression
    x = np.random.RandomState(1001).randn(512, 512).astype(np.float32)
    m = load_model('resnet8_u32')
    m.eval(); m.fill(); m.cuda()
    y = m(torch.from_numpy(x)[None, None].cuda())[0, 0]
    sd = {k: v.numpy() for k, v in m.state_dict().items()}
    ref = oscoring.score('resnet8', sd, x)
    assert np.abs(y.cpu().numpy() - ref).max() <= ATOL
    s, c = non_maximum_suppression(y, 8, threshold=-6.0)
    so, co = onms.nms2d(ref, 8, -6.0)
    assert len(s) > 100
    got, want = set(map(tuple, c.tolist())), set(map(tuple, co.tolist()))
    # a pick may differ only where two competing pixels are within the score tolerance
    for (px, py) in got ^ want:
        y0, y1, x0, x1 = max(0, py - 8), py + 9, max(0, px - 8), px + 9
        win = ref[y0:y1, x0:x1]
        assert np.sort(win.ravel())[-1] - ref[py, px] < 2 * ATOL, (px, py)
    assert len(got ^ want) <= max(2, len(want) // 100)
    # and NMS of the oracle's map on the device is bit-exact
    s2, c2 = non_maximum_suppression(ref, 8, threshold=-6.0)
    assert np.array_equal(c2, co) and np.array_equal(s2, so)


def test_patched_scoring_vs_reference_golden(gpu_ctx):
    from topaz_amd.model.factory import load_model
    from topaz_amd.model.utils import predict_in_patches
    z = load_golden('score_patched_resnet8_u32')
    m = load_model('resnet8_u32')
    m.eval(); m.fill(); m.cuda()
    patch = int(z['patch'])
    y = predict_in_patches(m, torch.from_numpy(z['x0'])[None, None], patch + 2 * (m.width // 2), use_cuda=True)
    assert y.dtype == np.float64 and y.shape == (1, 1) + z['y0'].shape
    assert np.abs(y[0, 0] - z['y0']).max() <= ATOL


@pytest.mark.parametrize('name', ['resnet8_bn_u16', 'conv127_bn_u16'])
def test_user_model_pickle(gpu_ctx, name):
    """full-module pickles written by the reference's torch.save(model) (training.py:601) load
    without the reference package installed and score like the reference did"""
    import os
    from conftest import GOLDEN
    from topaz_amd.model.factory import load_model
    z = load_golden(f'score_{name}')
    m = load_model(os.path.join(GOLDEN, f'user_model_{name}.sav'))
    assert np.abs(_score(m, z['x0']) - z['y0']).max() <= ATOL


@pytest.mark.parametrize('name', ['resnet8_drop_bn_u16', 'conv31_drop_bn_u16'])
def test_user_model_trained_with_dropout(gpu_ctx, name):
    """`topaz train --dropout p` pickles: nn.Dropout modules shift the module indices (resnet.py:296-303) and, in BasicConv,
    make fill() slip (basic.py:57-89: conv31 is scored with dilations 1,4,4) -- the golden is what the reference's own
    filled, eval-mode model returned"""
    import os
    from conftest import GOLDEN
    from topaz_amd.model.factory import load_model
    z = load_golden('score_dropout_models')
    m = load_model(os.path.join(GOLDEN, f'user_model_{name}.sav'))
    assert m.dropout
    assert np.abs(_score(m, z[name + ':x']) - z[name + ':y']).max() <= ATOL


def test_full_size_4096_windows_vs_oracle(gpu_ctx):
    """BASELINE size (4096x4096): the filled net is translation equivariant with a 71-pixel receptive
    field, so any window of the full-size score map must equal the oracle run on the window's own
    crop (+35 halo, zero beyond the image).  Checks interior, edge and corner windows."""
    from topaz_amd.model.factory import load_model
    x = np.random.RandomState(1000).randn(4096, 4096).astype(np.float32)
    m = load_model('resnet8_u32')
    y = _score(m, x)
    assert y.shape == (4096, 4096)
    sd = {k: v.numpy() for k, v in m.state_dict().items()}
    p = 35
    for (y0, x0) in ((0, 0), (1900, 2100), (3840, 0), (3840, 3840), (777, 3840)):
        h = w = 256
        ys, xs = max(0, y0 - p), max(0, x0 - p)
        ye, xe = min(4096, y0 + h + p), min(4096, x0 + w + p)
        crop = x[ys:ye, xs:xe]
        ref = oscoring.score('resnet8', sd, crop)
        # rows/cols of the crop that are at the image border carry the reference's zero padding; rows cut
        # inside the image are only valid `p` pixels away from the cut
        ref_win = ref[y0 - ys:y0 - ys + h, x0 - xs:x0 - xs + w]
        assert np.abs(y[y0:y0 + h, x0:x0 + w] - ref_win).max() <= ATOL, (y0, x0)


def test_edge_tiny_and_ragged_images(gpu_ctx):
    """images smaller than the 71-pixel receptive field / a single tile, and ragged (odd, prime) sizes"""
    from topaz_amd.model.factory import load_model
    m = load_model('resnet8_u32')
    sd = {k: v.numpy() for k, v in m.state_dict().items()}
    for shape in ((1, 1), (7, 5), (20, 33), (71, 71), (101, 257)):
        x = np.random.RandomState(sum(shape)).randn(*shape).astype(np.float32)
        y = _score(m, x)
        assert y.shape == shape
        assert np.abs(y - oscoring.score('resnet8', sd, x)).max() <= ATOL, shape


def test_batched_scoring_vs_oracle(gpu_ctx):
    """topaz.predict.batches / score_stream / score (predict.py:7-35): images are stacked batch_size at a time, the
    batch goes through model(x.unsqueeze(1)), one array per image comes back -- each must equal the ORACLE's logits
    of that image (a ragged last batch included), and a batch must not depend on its neighbours"""
    from topaz_amd.model.factory import load_model
    from topaz_amd.predict import batches, score, score_stream
    m = load_model('resnet16_u32')
    m.eval(); m.fill(); m.cuda()
    sd = {k: v.numpy() for k, v in m.state_dict().items()}
    imgs = [np.random.RandomState(s).randn(64, 80).astype(np.float32) for s in (1, 2, 3, 4, 5)]
    refs = [oscoring.score('resnet16', sd, x) for x in imgs]
    assert [tuple(b.shape) for b in batches(imgs, batch_size=2)] == [(2, 64, 80), (2, 64, 80), (1, 64, 80)]
    for bs in (1, 2, 5):
        out = score(m, imgs, use_cuda=True, batch_size=bs)
        assert len(out) == 5
        for y, ref in zip(out, refs):
            assert y.shape == ref.shape and np.abs(y - ref).max() <= ATOL
    streamed = list(score_stream(m, iter(imgs), use_cuda=True, batch_size=3))
    for y, y1 in zip(streamed, score(m, imgs, use_cuda=True, batch_size=1)):
        assert np.array_equal(y, y1)


# ---- 3-D scoring (`--dims 3`): filled ResNet8 / ResNet16 over Conv3d weights, classify_patches, NMS-3D -------------------
@pytest.mark.parametrize('name', ['resnet8_3d_u8', 'resnet8_3d_bn_u8', 'resnet16_3d_u8'])
def test_scoring_3d_vs_reference_golden(gpu_ctx, name):
    from topaz_amd.model.classifier import LinearClassifier
    z = load_golden(f'score_{name}')
    m = LinearClassifier(str(z['arch']), golden_sd(z))
    assert m.dims == 3 and m.width == {'resnet8': 71, 'resnet16': 91}[str(z['arch'])]
    m.eval(); m.fill(); m.cuda()
    x = z['x0']
    y = m(torch.from_numpy(x)[None, None].cuda())[0, 0].cpu().numpy()
    assert y.shape == x.shape
    assert np.abs(y - z['y0']).max() <= ATOL
    assert np.abs(y - oscoring.score(str(z['arch']), golden_sd(z), x)).max() <= ATOL
    # the default path is the plane-stacked 2xf16 one (dilated k^3 convs, 3-D residuals, fused head); the fp32 kernels agree
    eligible, split_runs, fp32_reruns = m.device_model.split_stats()
    assert eligible and split_runs == 1 and fp32_reruns == 0
    # ... and it is EVERY layer's path: a layer goes to the 2xf16 kernels when pick_split finds one, and for these networks it
    # finds one for each (include/topaz_hip.h states the rule; this pins which path the 3-D pickers take)
    n_conv, n_split, off = m.device_model.split_layers()
    assert n_conv > 0 and n_split == n_conv, off
    gpu_ctx.set_exact(True)
    try:
        y32 = m(torch.from_numpy(x)[None, None].cuda())[0, 0].cpu().numpy()
    finally:
        gpu_ctx.set_exact(False)
    assert np.abs(y32 - z['y0']).max() <= ATOL
    assert m.device_model.split_stats()[1] == 1


def test_scoring_3d_wide_net_against_float64(gpu_ctx):
    """a 3-D ResNet8 of 16 units (channels 16 .. 128: the 64- and 128-channel tiles, dilations 1 .. 4, the 5^3 head) on a 32^3
    volume: 2xf16 vs the float64 oracle next to torch's own fp32 error"""
    from topaz_amd.model.classifier import LinearClassifier
    sd = oscoring.synthetic_resnet_sd('resnet8', 16, 11, dims=3)
    m = LinearClassifier('resnet8', sd)
    assert m.dims == 3
    m.eval(); m.fill(); m.cuda()
    x = np.random.RandomState(5).randn(32, 32, 32).astype(np.float32)
    ref64 = oscoring.score('resnet8', sd, x, dtype=torch.float64)
    ref32 = oscoring.score('resnet8', sd, x)
    y = m(torch.from_numpy(x)[None, None].cuda())[0, 0].cpu().numpy()
    eligible, split_runs, fp32_reruns = m.device_model.split_stats()
    assert eligible and split_runs == 1 and fp32_reruns == 0
    e_split = float(np.abs(y.astype(np.float64) - ref64).max())
    e_t32 = float(np.abs(ref32.astype(np.float64) - ref64).max())
    print(f'resnet8-3d-u16 32^3 vs float64: max |2xf16| {e_split:.2e}  |torch fp32| {e_t32:.2e}')
    assert e_split <= ATOL and e_split <= 2.0 * max(e_t32, 1e-6)


def test_scoring_3d_volume_beyond_32bit_offsets_stays_on_fp32(gpu_ctx):
    """the plane-stacked 2xf16 kernels address one half of a split tensor with 32-bit byte offsets: a volume whose widest
    activation (128 channels = 16 cells here) would exceed 4 GiB per half -- more than 256^3 voxels -- is scored on the
    fp32 kernels instead of failing (rt_forward.hip split_volume_fits); one voxel fewer per plane and it takes the 2xf16 path"""
    from topaz_amd.model.classifier import LinearClassifier
    from tools import synth_weights as sw
    sd = sw.calibrate_head(sw.resnet_sd_uncalibrated('resnet8', 32, 3, dims=3), (1.0, 0.0))
    m = LinearClassifier('resnet8', sd)
    m.eval(); m.fill(); m.cuda()
    big = torch.from_numpy(np.random.RandomState(9).randn(258, 256, 256).astype(np.float32)).cuda()
    y = m(big[None, None])[0, 0]
    eligible, split_runs, fp32_reruns = m.device_model.split_stats()
    assert eligible and split_runs == 0 and fp32_reruns == 0          # never tried: not an overflow re-run
    gpu_ctx.set_exact(True)
    try:
        y32 = m(big[None, None])[0, 0]
    finally:
        gpu_ctx.set_exact(False)
    assert torch.equal(y, y32) and bool(torch.isfinite(y).all())
    del y, y32, big
    small = torch.from_numpy(np.random.RandomState(9).randn(64, 256, 256).astype(np.float32)).cuda()
    m(small[None, None])
    assert m.device_model.split_stats()[1] == 1


def test_scoring_3d_patches_nms_and_user_pickle(gpu_ctx):
    """classify_patches (PatchDataset tiles, zero-filled halo) vs the reference's output; the 3-D pick table of the
    reference's own score map; a full-module pickle of a 3-D classifier loads and scores like the reference did"""
    import os
    from conftest import GOLDEN
    from topaz_amd.algorithms import non_maximum_suppression_3d
    from topaz_amd.model.classifier import classify_patches
    from topaz_amd.model.factory import load_model
    z = load_golden('score_resnet8_3d_u8')
    m = load_model(os.path.join(GOLDEN, 'user_model_resnet8_3d_u8.sav'))
    assert m.dims == 3
    m.eval(); m.fill(); m.cuda()
    x = z['x0']
    y = m(torch.from_numpy(x)[None, None].cuda())[0, 0].cpu().numpy()
    assert np.abs(y - z['y0']).max() <= ATOL
    yp = classify_patches(m, torch.from_numpy(x)[None], patch_size=int(z['patch_size']), padding=int(z['padding']),
                          verbose=False)[0].numpy()
    assert np.abs(yp - z['patched']).max() <= ATOL
    s, c = non_maximum_suppression_3d(z['y0'], int(z['nms_r']), threshold=float(z['nms_thr']))
    assert np.array_equal(c, z['nms_coords']) and np.array_equal(s, z['nms_scores'])


@pytest.mark.parametrize('name', ['conv31_3d_bn_u8', 'conv63_3d_bn_u8', 'conv127_3d_bn_u8'])
def test_scoring_3d_basicconv_stacks_vs_reference_golden(gpu_ctx, name):
    """3-D BasicConv stacks (basic.py:12-111 with dims = 3: `-m conv31 --dims 3`): 7^3 stem, dilated 5^3 convs with folded
    eval-BN and per-layer PReLU slopes, fused 1x1x1 head -- against the reference's own output; conv31 also from its
    full-module pickle.  conv127 (round 6; until then refused): its last 5^3 conv at dilation 16 reads 64 columns either side of
    every row of every plane, too much for the LDS tiles of the 2xf16 kernels -- that one layer runs on a small-tile fp32-MFMA
    instantiation (2 x 2 x 16 outputs per workgroup: slow, correct), the others where conv63's run"""
    import os
    from conftest import GOLDEN
    from topaz_amd.model.classifier import LinearClassifier
    from topaz_amd.model.factory import load_model
    z = load_golden(f'score_{name}')
    arch = str(z['arch'])
    m = LinearClassifier(arch, golden_sd(z))
    assert m.dims == 3 and m.width == {'conv31': 31, 'conv63': 63, 'conv127': 127}[arch]
    m.eval(); m.fill(); m.cuda()
    x = z['x0']
    y = m(torch.from_numpy(x)[None, None].cuda())[0, 0].cpu().numpy()
    assert y.shape == x.shape
    assert np.abs(y - z['y0']).max() <= ATOL
    assert np.abs(y - oscoring.score(arch, golden_sd(z), x)).max() <= ATOL
    if arch == 'conv31':
        m2 = load_model(os.path.join(GOLDEN, 'user_model_conv31_3d_bn_u8.sav'))
        assert m2.dims == 3
        m2.eval(); m2.fill(); m2.cuda()
        assert np.abs(m2(torch.from_numpy(x)[None, None].cuda())[0, 0].cpu().numpy() - z['y0']).max() <= ATOL


# ---- ResNets with MaxPool layers: ResNet6 and `topaz train --pooling max` (resnet.py:10-47,254-339) ----------------------
@pytest.mark.parametrize('name', ['resnet6_u16', 'resnet8_pool_bn_u16', 'resnet16_pool_u8'])
def test_pooled_resnets_vs_reference_golden(gpu_ctx, name):
    """filled MaxPool(3, stride 2) = a 3x3 max at the accumulated dilation, stride 1 (TPZ_OP_MAXPOOL), on fp32 planes or
    split cells depending on its neighbours"""
    from topaz_amd.model.classifier import LinearClassifier
    z = load_golden(f'score_{name}')
    m = LinearClassifier(str(z['arch']), golden_sd(z), pooling=True)
    assert m.width == int(z['width']) and m.fill() == 4
    y = _score(m, z['x0'])
    n_conv, n_split, off = m.device_model.split_layers()
    assert n_split == n_conv, off              # (ResNet6's 5x5 stem has a 5 x 1 column kernel: no layer on an fp32 kernel)
    assert y.shape == z['y0'].shape
    assert np.abs(y - z['y0']).max() <= ATOL
    gpu_ctx.set_exact(True)
    try:
        y32 = _score(m, z['x0'])
    finally:
        gpu_ctx.set_exact(False)
    assert np.abs(y32 - z['y0']).max() <= ATOL


def test_pooled_resnet_user_pickle_and_larger_image(gpu_ctx):
    import os
    from conftest import GOLDEN
    from topaz_amd.model.factory import load_model
    z = load_golden('score_resnet8_pool_bn_u16')
    m = load_model(os.path.join(GOLDEN, 'user_model_resnet8_pool_bn_u16.sav'))
    assert m.pooling and m.arch == 'resnet8' and m.width == 77
    assert np.abs(_score(m, z['x0']) - z['y0']).max() <= ATOL
    x = np.random.RandomState(77).randn(333, 290).astype(np.float32)
    sd = {k: v.numpy() for k, v in m.state_dict().items()}
    assert np.abs(_score(m, x) - oscoring.score('resnet8', sd, x, pooling=True)).max() <= ATOL


@pytest.mark.parametrize('exact', [False, True], ids=['2xf16', 'exact_fp32'])
def test_internal_tiling_is_bit_identical(gpu_ctx, exact):
    """`topaz extract` scores any image that fits memory (extract.py:247-249); above the tiling limit tpz_model_forward scores a
    2-D image tile by tile (each tile grown by the network's receptive halo) instead of refusing it for its 32-bit cell
    offsets.  Forced here on images that also fit whole: identical bits -- the pretrained ResNet8 and a 64-unit one, tiles that
    do not divide the image, a raw-count image (one range exponent for the whole image), both arithmetic paths."""
    from topaz_amd.model.classifier import LinearClassifier
    from topaz_amd.model.factory import load_model
    nets = [load_model('resnet8_u32'), LinearClassifier('resnet8', oscoring.synthetic_resnet_sd('resnet8', 64, 7))]
    rs = np.random.RandomState(5)
    try:
        gpu_ctx.set_exact(exact)
        for m in nets:
            for shape, tile, raw in (((300, 420), 96, False), ((257, 199), 128, True)):
                x = rs.randn(*shape).astype(np.float32)
                if raw:
                    x = np.round(x * 300 + 2500).astype(np.float32)
                gpu_ctx.set_tiling(40 << 20, 4096)
                whole = _score(m, x)
                gpu_ctx.set_tiling(1, tile)
                tiled = _score(m, x)
                assert whole.shape == x.shape and np.isfinite(whole).all()
                assert np.array_equal(whole, tiled), (shape, tile, raw)
    finally:
        gpu_ctx.set_tiling(40 << 20, 4096)
        gpu_ctx.set_exact(False)


def test_large_frame_is_scored_in_tiles(gpu_ctx):
    """a frame above the default tiling limit (here 7000 x 6200 = 43.4 Mpx through the pretrained ResNet8): sampled windows
    agree with the oracle run on the window's own crop (translation equivariance away from the borders)"""
    from topaz_amd.model.factory import load_model
    m = load_model('resnet8_u32')
    H, W = 7000, 6200
    x = np.random.RandomState(9).randn(H, W).astype(np.float32)
    y = _score(m, x)
    assert y.shape == (H, W)
    sd = {k: v.numpy() for k, v in m.state_dict().items()}
    R = 71                                                   # (width of the filled ResNet8: the oracle crop's own halo)
    for (cy, cx) in ((4096, 4096), (100, 6100), (6900, 3000)):
        y0, x0 = max(0, cy - 64 - R), max(0, cx - 64 - R)
        y1, x1 = min(H, cy + 64 + R), min(W, cx + 64 + R)
        ref = oscoring.score('resnet8', sd, x[y0:y1, x0:x1].copy())
        a, b = (R if y0 > 0 else 0), (R if x0 > 0 else 0)
        c, d = (ref.shape[0] - R if y1 < H else ref.shape[0]), (ref.shape[1] - R if x1 < W else ref.shape[1])
        assert np.abs(y[y0 + a:y0 + c, x0 + b:x0 + d] - ref[a:c, b:d]).max() <= 1e-4


@pytest.mark.parametrize('name', ['resnet8_u32', 'resnet16_u32', 'resnet8_bn_u32'])
def test_weights_resident_kernel_for_the_32_unit_detectors(gpu_ctx, name):
    """The 3x3 32 -> 32 ResidA layers (resnet.py:108-204 filled: dilation 1 / 2 / 4, plain / residual / residual + eval-BN; nine
    of resnet16_u32's 17 layers) run on a persistent kernel that keeps the layer's weights in the LDS (csrc/conv_rw.h: no
    per-step DMA, plan or barrier) instead of the general 2xf16 tile.  Image sizes that do not divide the 8 x 32 tile, the
    CLI's raw-count case through range scaling, the fp32 oracle to 1e-4 both ways, the two kernels against each other to
    rounding (another summation order), and the launches really are the new kernel's."""
    from topaz_amd.model.classifier import LinearClassifier
    from topaz_amd.model.factory import load_model
    if name.endswith('_bn_u32'):
        sd = oscoring.synthetic_resnet_sd('resnet8', 32, 5, bn=True)
        m, arch = LinearClassifier('resnet8', sd), 'resnet8'
    else:
        m, arch = load_model(name), name.split('_')[0]
        sd = {k: v.numpy() for k, v in m.state_dict().items()}
    rs = np.random.RandomState(64)
    for shape, gain, offset in (((333, 401), 1.0, 0.0), ((512, 640), 400.0, 3000.0)):
        x = (rs.randn(*shape) * gain + offset).astype(np.float32)
        ref = oscoring.score(arch, sd, x)
        tol = max(ATOL, 3e-6 * float(np.abs(ref).max()))
        gpu_ctx.prof_enable(1); gpu_ctx.prof_reset()
        try:
            y = _score(m, x)
            names_on = [k[0] for k in gpu_ctx.prof_kernels()]
            gpu_ctx.set_rw(False)
            gpu_ctx.prof_reset()
            y_gen = _score(m, x)
            names_off = [k[0] for k in gpu_ctx.prof_kernels()]
        finally:
            gpu_ctx.set_rw(True)
            gpu_ctx.prof_enable(False)
        assert any('conv_split_rw_kernel' in n for n in names_on) and not any('conv_split_rw_kernel' in n for n in names_off)
        e_rw, e_gen, e_ab = np.abs(y - ref).max(), np.abs(y_gen - ref).max(), np.abs(y - y_gen).max()
        print(f'{name} {shape} x*{gain}+{offset}: |rw - oracle| {e_rw:.2e}  |general - oracle| {e_gen:.2e}  |rw - general| {e_ab:.2e}')
        assert e_rw <= tol and e_gen <= tol and e_ab <= tol


def test_partial_program_of_odd_width_ending_in_a_pool_keeps_its_channels(gpu_ctx):
    """A program whose widths have no 2xf16 kernel is loaded zero-padded to multiples of 16 channels (widen_program); the
    tensor the program RETURNS keeps its width also when the program ends in a pool behind the padded conv (round-5 advisor
    finding: the caller got 16 channels, four of them zero, and the shape depended on which kernels exist)."""
    import torch.nn.functional as F
    from topaz_amd.runtime import DeviceModel, LayerProgram
    rs = np.random.RandomState(5)
    w0 = (rs.randn(12, 1, 3, 3) * 0.3).astype(np.float32)
    w1 = (rs.randn(12, 12, 3, 3) * 0.15).astype(np.float32)
    b0, b1 = (rs.randn(12) * 0.1).astype(np.float32), (rs.randn(12) * 0.1).astype(np.float32)
    x = rs.randn(1, 1, 70, 90).astype(np.float32)
    t = F.leaky_relu(F.conv2d(torch.from_numpy(x), torch.from_numpy(w0), torch.from_numpy(b0), padding=1), 0.1)
    t = F.leaky_relu(F.conv2d(t, torch.from_numpy(w1), torch.from_numpy(b1), padding=1), 0.1)
    for tail, ref in (('maxpool2', F.max_pool2d(t, 2)), ('maxpool', F.max_pool2d(t, 3, stride=1))):
        p = LayerProgram(2)
        s = p.conv(0, w0, b0, pad=1, slope=0.1)
        s = p.conv(s, w1, b1, pad=1, slope=0.1)
        s = p.maxpool2(s) if tail == 'maxpool2' else p.maxpool(s, 3, 1)
        y = DeviceModel(p, gpu_ctx).forward(torch.from_numpy(x).cuda()).cpu()
        assert tuple(y.shape) == tuple(ref.shape), (tail, y.shape, ref.shape)
        assert (y - ref).abs().max().item() <= ATOL
