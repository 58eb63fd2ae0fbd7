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
