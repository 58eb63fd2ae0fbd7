import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs an MI355X (gfx950) and the built libtopaz_hip.so')


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)


def golden_sd(z, prefix='sd:'):
    return {k[len(prefix):]: z[k] for k in z.files if k.startswith(prefix)}


@pytest.fixture(scope='session')
def gpu_ctx():
    """tpz context on cuda:0; GPU tests must FAIL (not skip) when the library cannot be used"""
    import torch
    assert torch.cuda.is_available(), 'gpu-marked test without a visible GPU'
    from topaz_amd.runtime import get_context
    return get_context(0)
