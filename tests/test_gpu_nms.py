"""HIP NMS (tpz_nms_2d/3d) must be bit-identical to the reference's greedy loop: the golden
known-answer cases, and the C oracle on seeded maps (ties, edges, big radii)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import nms as onms

pytestmark = pytest.mark.gpu


def _run(x, r, thr, scale=1.0):
    from topaz_amd import runtime as rt
    s, c = rt.nms(torch.from_numpy(np.ascontiguousarray(x)), r, thr, scale=scale)
    return s.cpu().numpy(), c.cpu().numpy()


def test_golden_cases(gpu_ctx):
    z = load_golden('nms_cases')
    names = sorted({k.split(':')[0] for k in z.files if ':' in k})
    for name in names:
        x, r, thr = z[name + ':x'], int(z[name + ':r']), float(z[name + ':thr'])
        scale = float(z[name + ':scale']) if x.ndim == 3 else 1.0
        s, c = _run(x, r, thr, scale)
        assert np.array_equal(c, z[name + ':coords']), name
        assert np.array_equal(s, z[name + ':scores']), name


@pytest.mark.parametrize('shape,r,thr', [((257, 300), 5, -0.3), ((128, 64), 14, -1.0), ((33, 500), 8, 0.2),
                                         ((5, 7), 9, -9.0), ((64, 64), 0, 0.0), ((100, 90), 3, 5.0)])
def test_random_maps_vs_oracle(gpu_ctx, shape, r, thr):
    rs = np.random.RandomState(shape[0] * 7 + r)
    x = rs.randn(*shape).astype(np.float32)
    x[rs.rand(*shape) < 0.05] = 0.5           # exact ties
    x[rs.rand(*shape) < 0.01] = -0.0
    so, co = onms.nms2d(x, r, thr)
    s, c = _run(x, r, thr)
    assert np.array_equal(c, co) and np.array_equal(s, so)


def test_ramp_long_dependency_chain(gpu_ctx):
    # monotone ramp: every pixel depends on its higher neighbour -> many relaxation sweeps
    x = (np.arange(40 * 400, dtype=np.float32).reshape(40, 400)) / 100.0
    so, co = onms.nms2d(x, 3, 1.0)
    s, c = _run(x, 3, 1.0)
    assert np.array_equal(c, co) and np.array_equal(s, so)


@pytest.mark.parametrize('shape,r,scale,thr', [((20, 24, 28), 2, 1.0, 0.5), ((9, 40, 33), 3, 1.5, -0.2),
                                               ((4, 5, 6), 2, 2.0, -5.0)])
def test_random_volumes_vs_oracle(gpu_ctx, shape, r, scale, thr):
    rs = np.random.RandomState(shape[0] + r)
    v = rs.randn(*shape).astype(np.float32)
    v[rs.rand(*shape) < 0.05] = 0.75
    so, co = onms.nms3d(v, r, scale, thr)
    s, c = _run(v, r, thr, scale)
    assert np.array_equal(c, co) and np.array_equal(s, so)


def test_large_map_properties(gpu_ctx):
    """4096^2 (BASELINE size): size-independent properties + full comparison with the C oracle."""
    rs = np.random.RandomState(1000)
    x = (rs.randn(4096, 4096) * 2.5 - 7.0).astype(np.float32)
    r, thr = 14, -6.0
    s, c = _run(x, r, thr)
    assert len(s) > 1000
    assert np.all(s[:-1] >= s[1:]) and np.all(s > thr)                  # sortedness, threshold
    assert np.array_equal(x[c[:, 1], c[:, 0]], s)                        # scores are the map values
    # idempotence: NMS of a map holding only the picks returns the same picks
    y = np.full_like(x, -100.0)
    y[c[:, 1], c[:, 0]] = s
    s2, c2 = _run(y, r, thr)
    assert np.array_equal(c2, c) and np.array_equal(s2, s)
    so, co = onms.nms2d(x, r, thr)
    assert np.array_equal(c, co) and np.array_equal(s, so)


def test_randomised_2d_sweep_vs_oracle(gpu_ctx):
    """40 random configurations (shape, radius, threshold, tie density, plateaus, NaN / inf entries)"""
    rs = np.random.RandomState(12345)
    for trial in range(40):
        H, W = int(rs.randint(1, 200)), int(rs.randint(1, 260))
        r = int(rs.choice([0, 1, 2, 3, 5, 8, 14, 20]))
        x = rs.randn(H, W).astype(np.float32)
        kind = trial % 5
        if kind == 1:
            x = np.round(x * 2) / 2                        # heavy ties
        elif kind == 2:
            x[:] = 1.0                                     # one plateau: pure tie-break order
        elif kind == 3 and H * W > 4:
            idx = rs.randint(0, H * W, size=3)
            x.ravel()[idx[0]] = np.inf
            x.ravel()[idx[1]] = -np.inf
        elif kind == 4:
            x = (x * 1e-3).astype(np.float32)              # dense near-threshold values
        thr = float(rs.choice([-np.inf, -1.0, 0.0, 0.5]))
        so, co = onms.nms2d(x, r, thr)
        s, c = _run(x, r, thr)
        assert np.array_equal(c, co), (trial, H, W, r, thr)
        assert np.array_equal(s, so), (trial, H, W, r, thr)


def test_randomised_3d_sweep_vs_oracle(gpu_ctx):
    rs = np.random.RandomState(54321)
    for trial in range(15):
        D, H, W = int(rs.randint(1, 20)), int(rs.randint(1, 30)), int(rs.randint(1, 40))
        r, scale = int(rs.choice([0, 1, 2, 3])), float(rs.choice([1.0, 1.5, 2.0]))
        v = rs.randn(D, H, W).astype(np.float32)
        if trial % 3 == 1:
            v = np.round(v)
        thr = float(rs.choice([-np.inf, -0.5, 0.3]))
        so, co = onms.nms3d(v, r, scale, thr)
        s, c = _run(v, r, thr, scale)
        assert np.array_equal(c, co) and np.array_equal(s, so), (trial, D, H, W, r, scale, thr)


def test_nan_scores_sort_first(gpu_ctx):
    """numpy's argsort puts NaN last, so reversed they are visited first; `NaN <= threshold` is False, so a
    NaN pixel is a pick (and suppresses its neighbourhood) -- for either sign of the NaN payload"""
    rs = np.random.RandomState(9)
    x = rs.randn(40, 50).astype(np.float32)
    x[5, 7] = np.nan
    x[20, 30] = np.float32(np.nan) * -1
    x.view(np.uint32)[30, 10] = 0xFFC00001              # negative quiet NaN with a payload
    so, co = onms.nms2d(x, 4, 0.0)
    s, c = _run(x, 4, 0.0)
    assert np.array_equal(c, co)
    assert np.array_equal(np.isnan(s), np.isnan(so)) and np.array_equal(s[~np.isnan(s)], so[~np.isnan(so)])
    assert np.isnan(s[:3]).all() and set(map(tuple, c[:3].tolist())) == {(7, 5), (30, 20), (10, 30)}


def test_patched_suppression_matches_whole_image_on_separated_peaks(gpu_ctx):
    """NonMaximumSuppression with tiles (the branch that cannot run upstream, extract.py:42-72): peaks further apart than
    the suppression radius and an overlap >= radius make tile-wise and whole-image suppression agree, borders included
    (2-D and 3-D)"""
    from topaz_amd.extract import NonMaximumSuppression
    rs = np.random.RandomState(9)
    x = np.full((150, 170), -10.0, dtype=np.float32)
    ys, xs = np.meshgrid(np.arange(3, 150, 12), np.arange(2, 170, 12), indexing='ij')
    x[ys, xs] = rs.rand(*ys.shape).astype(np.float32) + 1
    whole = NonMaximumSuppression(5, -6.0, patch_size=0)(('a', x))
    tiled = NonMaximumSuppression(5, -6.0, patch_size=64, patch_overlap=8)(('a', x))
    assert len(whole[1]) == ys.size
    assert sorted(map(tuple, whole[2].tolist())) == sorted(map(tuple, tiled[2].tolist()))
    assert sorted(whole[1].tolist()) == sorted(tiled[1].tolist())
    v = np.full((40, 44, 48), -10.0, dtype=np.float32)
    zz, yy, xx = np.meshgrid(np.arange(2, 40, 9), np.arange(3, 44, 9), np.arange(1, 48, 9), indexing='ij')
    v[zz, yy, xx] = rs.rand(*zz.shape).astype(np.float32) + 1
    whole = NonMaximumSuppression(3, -6.0, dims=3, patch_size=0)(('v', v))
    tiled = NonMaximumSuppression(3, -6.0, dims=3, patch_size=24, patch_overlap=4)(('v', v))
    assert len(whole[1]) == zz.size
    assert sorted(map(tuple, whole[2].tolist())) == sorted(map(tuple, tiled[2].tolist()))
    with pytest.raises(ValueError, match='no core'):
        NonMaximumSuppression(5, -6.0)(('a', x))           # the upstream default 64 / 32 has step 0
