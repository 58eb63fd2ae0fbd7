"""CPU-only checks of the host side: the C-ABI library loads and exports every declared symbol,
the ctypes struct matches the header, the packers emit the right graphs, model files parse."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, golden_sd, load_golden


def test_library_exports_every_declared_symbol():
    from topaz_amd import _lib
    header = open(os.path.join(ROOT, 'include', 'topaz_hip.h')).read()
    declared = set(re.findall(r'\b(tpz_[a-z0-9_]+)\s*\(', header))
    assert declared, 'no prototypes found in the header'
    lib = C.CDLL(_lib.LIB_PATH)                       # raw dlopen: no GPU needed
    for name in sorted(declared):
        assert hasattr(lib, name), f'{name} declared in include/topaz_hip.h but not exported'
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
    _lib.load_library()                               # prototypes attach without error


def test_stray_debug_variables_change_nothing_without_TPZ_DEBUG(monkeypatch):
    """The library's A/B switches (TPZ_EXACT_FP32, TPZ_NO_ROI, ...) are read by ONE function and only under TPZ_DEBUG=1
    (VERDICT r05 item 7): a variable left in a user's environment must not change the arithmetic or the schedule of a job."""
    import ctypes as C
    from topaz_amd import _lib
    lib = _lib.load_library()
    buf = C.create_string_buffer(512)
    for k in ('TPZ_DEBUG', 'TPZ_EXACT_FP32', 'TPZ_NO_ROI', 'TPZ_NO_RANGE', 'TPZ_NO_WIDEN', 'TPZ_BATCH', 'TPZ_LANES'):
        monkeypatch.delenv(k, raising=False)
    assert lib.tpz_debug_switches(buf, 512) == 0 and buf.value == b''
    for k, v in (('TPZ_EXACT_FP32', '1'), ('TPZ_NO_ROI', '1'), ('TPZ_NO_RANGE', '1'), ('TPZ_NO_WIDEN', '1'), ('TPZ_BATCH', '2'),
                 ('TPZ_LANES', '3')):
        monkeypatch.setenv(k, v)
    assert lib.tpz_debug_switches(buf, 512) == 0 and buf.value == b'', buf.value      # stray variables: nothing in effect
    monkeypatch.setenv('TPZ_DEBUG', '0')
    assert lib.tpz_debug_switches(buf, 512) == 0
    monkeypatch.setenv('TPZ_DEBUG', '1')
    assert lib.tpz_debug_switches(buf, 512) == 6
    assert buf.value.split() == [b'exact_fp32', b'no_roi', b'no_range', b'no_widen', b'batch', b'lanes']
    # and no other getenv in the device library's sources
    import glob, os, re
    root = os.path.dirname(os.path.abspath(_lib.__file__))
    uses = []
    for f in glob.glob(os.path.join(root, 'csrc', '*.h*')):
        src = open(f).read()
        if not f.endswith('rt_core.hip'):
            assert 'getenv' not in src, f
        else:
            body = src[src.index('DebugEnv debug_env() {'):]
            body = body[:body.index('\n}\n') + 3]
            assert src.count('getenv') == body.count('getenv'), 'getenv outside debug_env()'


def test_layer_struct_matches_header():
    from topaz_amd._lib import TpzLayer
    header = open(os.path.join(ROOT, 'include', 'topaz_hip.h')).read()
    body = header[header.index('typedef struct tpz_layer {'):header.index('} tpz_layer;')]
    fields = re.findall(r'^\s*(int32_t|int64_t|float)\s+([a-z0-9_]+);', body, flags=re.M)
    ctype = {'int32_t': C.c_int32, 'int64_t': C.c_int64, 'float': C.c_float}
    assert [(n, ctype[t]) for t, n in fields] == list(TpzLayer._fields_)
    assert C.sizeof(TpzLayer) % 8 == 0


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is visible')
    from topaz_amd import runtime, _lib
    with pytest.raises(_lib.TopazHipError):
        runtime.get_context(0)
    lib = _lib.load_library()
    h = C.c_void_p()
    assert lib.tpz_ctx_create(0, C.byref(h)) != 0
    assert b'no HIP device' in lib.tpz_last_error(None)


def test_resnet_packer_graph():
    from topaz_amd.model import pack
    from oracle.scoring import synthetic_resnet_sd
    P, width = pack.pack_resnet('resnet8', synthetic_resnet_sd('resnet8', 32, 1))
    assert width == 71
    convs = [(L.k, L.dil, L.cin, L.cout, L.pad, L.res_crop, L.head) for L in P.layers]
    assert convs == [(7, 1, 1, 32, 35, 0, 0),
                     (3, 2, 32, 32, 0, 0, 0), (3, 4, 32, 32, 0, 6, 0),
                     (3, 2, 32, 32, 0, 0, 0), (1, 1, 32, 64, 0, 0, 0), (3, 4, 32, 64, 0, 6, 0),
                     (3, 4, 64, 64, 0, 0, 0), (3, 8, 64, 64, 0, 12, 0),
                     (5, 4, 64, 128, 0, 0, 1)]
    P16, w16 = pack.pack_resnet('resnet16', synthetic_resnet_sd('resnet16', 16, 2, bn=True))
    assert w16 == 91
    dils = [L.dil for L in P16.layers if L.k == 3]
    assert dils == [1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 4, 4, 4, 4]
    assert sum(1 for L in P16.layers if L.post_scale_off >= 0) == 7     # bn1 kept as an epilogue affine


def test_bn_folding_matches_oracle_arithmetic():
    from topaz_amd.model import pack
    rs = np.random.RandomState(0)
    w, b = rs.randn(8, 4, 3, 3).astype(np.float32), None
    sd = {'bn.weight': rs.rand(8).astype(np.float32) + .5, 'bn.bias': rs.randn(8).astype(np.float32),
          'bn.running_mean': rs.randn(8).astype(np.float32), 'bn.running_var': rs.rand(8).astype(np.float32) + .5}
    w2, b2 = pack._fold_bn(w, b, sd, 'bn')
    import torch
    import torch.nn.functional as F
    x = torch.randn(1, 4, 9, 9)
    ref = F.batch_norm(F.conv2d(x, torch.from_numpy(w)), torch.from_numpy(sd['bn.running_mean']),
                       torch.from_numpy(sd['bn.running_var']), torch.from_numpy(sd['bn.weight']),
                       torch.from_numpy(sd['bn.bias']), False, 0.0, 1e-5)
    got = F.conv2d(x, torch.from_numpy(w2), torch.from_numpy(b2))
    assert (ref - got).abs().max() < 1e-5


def test_unet_packer_graph():
    from topaz_amd.model import pack
    from oracle.denoising import synthetic_unet_sd
    P = pack.pack_unet(synthetic_unet_sd(1, nf=48, base_width=11, top_width=5), 5, 2)
    convs = [(L.cin, L.cout, L.k, L.src2 >= 0) for L in P.layers if L.op == 1]
    assert convs[0] == (1, 48, 11, False) and convs[-1] == (32, 1, 5, False)
    assert [c for c in convs if c[3]] == [(96, 96, 3, True), (144, 96, 3, True), (144, 96, 3, True),
                                          (144, 96, 3, True), (97, 64, 5, True)]
    assert sum(1 for L in P.layers if L.op == 2) == 5


@pytest.mark.parametrize('name,arch', [('resnet8_bn_u16', 'resnet8'), ('conv127_bn_u16', 'conv127')])
def test_user_model_pickle_parses_without_reference(name, arch):
    import sys
    assert 'topaz' not in sys.modules
    from topaz_amd.model.unpickle import load_module_pickle
    a, sd = load_module_pickle(os.path.join(GOLDEN, f'user_model_{name}.sav'))
    assert a == arch
    want = golden_sd(load_golden(f'score_{name}'))
    assert set(sd) == set(want)
    for k in want:
        assert np.array_equal(sd[k].numpy(), want[k]), k


def test_dropout_pickles_are_renumbered_and_basicconv_fill_slips():
    """models trained with --dropout: the unpickler drops the nn.Dropout indices; the packer reproduces upstream's fill()
    of such a BasicConv (basic.py:57-89) -- and the oracle, run on upstream's numbering, matches the reference's scores"""
    from oracle import scoring
    from topaz_amd.model import pack
    from topaz_amd.model.unpickle import load_module_pickle
    assert pack.basic_fill_dilations(3, True, False) == [1, 2, 4]
    assert pack.basic_fill_dilations(5, False, False) == [1, 2, 4, 8, 16]
    assert pack.basic_fill_dilations(3, True, True) == [1, 4, 4]
    z = load_golden('score_dropout_models')
    for name, arch in (('resnet8_drop_bn_u16', 'resnet8'), ('conv31_drop_bn_u16', 'conv31')):
        a, sd, traits = load_module_pickle(os.path.join(GOLDEN, f'user_model_{name}.sav'), with_traits=True)
        assert a == arch and traits == {'pooling': False, 'dropout': True}
        idx = sorted({int(k.split('.')[2]) for k in sd if k.startswith('features.features.')})
        assert idx == list(range(len(idx)))                       # contiguous: no holes where the Dropouts were
        # upstream's numbering back again -> the oracle's dropout=True walk
        kinds = {'resnet8': 'BDRRDRBD', 'conv31': 'CNADCNADCNAD'}[arch]
        keep = [i for i, c in enumerate(kinds) if c != 'D']
        raw = {}
        for k, v in sd.items():
            if k.startswith('features.features.'):
                parts = k.split('.')
                parts[2] = str(keep[int(parts[2])])
                k = '.'.join(parts)
            raw[k] = v.numpy()
        y = scoring.score(arch, raw, z[name + ':x'], dropout=True)
        assert np.abs(y - z[name + ':y']).max() < 3e-5


def test_patch_geometry_helpers():
    import torch
    from topaz_amd.model.utils import get_patches, reconstruct_from_patches, insize_from_outsize
    X = torch.arange(1, 1 + 50 * 70, dtype=torch.float32).reshape(1, 1, 50, 70)
    pad, size = 5, 30
    patches = get_patches(X, size, pad)
    assert len(patches) == 3 * 4 and patches[0].shape[-2:] == (30, 30)
    cropped = [p[0, 0, pad:-pad, pad:-pad].numpy() for p in patches]
    assert np.array_equal(reconstruct_from_patches(cropped, X.shape, size, pad)[0, 0], X[0, 0].numpy())
    assert insize_from_outsize([dict(kernel_size=7, stride=2), dict(kernel_size=5)], 1) == 15


def test_bench_weights_equal_test_weights():
    """bench.py / tools take their seeded ResNet weights from tools/synth_weights.py with recorded probe statistics
    (no oracle call); they must stay bit-identical to the oracle-calibrated weights the parity tests use."""
    import numpy as np
    from oracle import scoring as oscoring
    from tools import synth_weights as sw
    for (arch, units, seed, bn), stats in sw.KNOWN_PROBE_STATS.items():
        a = oscoring.synthetic_resnet_sd(arch, units, seed, bn)
        b = sw.calibrate_head(sw.resnet_sd_uncalibrated(arch, units, seed, bn), stats)
        assert list(a) == list(b)
        for k in a:
            assert np.array_equal(a[k], b[k]), (arch, k)


def test_unpickler_allowlist_refuses_foreign_globals(tmp_path):
    """a model file is data: globals outside the allowlist (here os.system via __reduce__) raise UnpicklingError instead
    of being called; the reference's own full-module pickles still load (tests/golden/user_model_*.sav)"""
    import os
    import pickle
    import torch
    from conftest import GOLDEN
    from topaz_amd.model.unpickle import _PickleModule, load_module_pickle

    class Evil:
        def __reduce__(self):
            return (os.system, ('echo pwned > ' + str(tmp_path / 'pwned'),))

    bad = tmp_path / 'evil.sav'
    torch.save({'w': torch.zeros(2), 'x': Evil()}, str(bad))
    with pytest.raises(pickle.UnpicklingError, match='system'):
        torch.load(str(bad), map_location='cpu', weights_only=False, pickle_module=_PickleModule)
    assert not (tmp_path / 'pwned').exists()
    arch, sd = load_module_pickle(os.path.join(GOLDEN, 'user_model_resnet8_bn_u16.sav'))
    assert arch == 'resnet8' and 'classifier.weight' in sd
    arch, sd = load_module_pickle(os.path.join(GOLDEN, 'user_model_conv127_bn_u16.sav'))
    assert arch == 'conv127'


def test_unpickler_nested_payload_stays_inside_the_allowlist(tmp_path):
    """torch.storage._load_from_bytes is `torch.load(BytesIO(b), weights_only=False)` with the STANDARD unpickler: a model
    file could wrap any payload in it.  The name is mapped onto a wrapper that re-enters this module's own
    unpickler: the nested os.system is refused, a nested tensor still loads."""
    import io
    import os
    import pickle
    import torch
    from topaz_amd.model.unpickle import _PickleModule

    class Evil:
        def __reduce__(self):
            return (os.system, ('echo pwned > ' + str(tmp_path / 'pwned2'),))

    class Nested:
        def __init__(self, payload):
            buf = io.BytesIO()
            torch.save(payload, buf)
            self.b = buf.getvalue()

        def __reduce__(self):
            return (torch.storage._load_from_bytes, (self.b,))

    bad = tmp_path / 'nested_evil.sav'
    torch.save({'x': Nested(Evil())}, str(bad), pickle_protocol=4)
    with pytest.raises(pickle.UnpicklingError, match='system'):
        torch.load(str(bad), map_location='cpu', weights_only=False, pickle_module=_PickleModule)
    assert not (tmp_path / 'pwned2').exists()
    ok = tmp_path / 'nested_ok.sav'
    torch.save({'x': Nested(torch.arange(5.))}, str(ok), pickle_protocol=4)
    got = torch.load(str(ok), map_location='cpu', weights_only=False, pickle_module=_PickleModule)
    assert torch.equal(got['x'], torch.arange(5.))


def test_denoiser_pickles_dispatch_on_the_class_name(tmp_path):
    """UDenoiseNet3 (--arch unet3) has UDenoiseNet's parameter names but returns x - dec1(h): the pickled CLASS decides which
    forward pass is built, never the parameter names alone; DenoiseNet (--arch fcnet), whose forward pass
    raises upstream, and unknown classes are refused.  The classes are faked under the reference's module path for pickling."""
    import sys
    import types
    import torch
    from tools import synth_weights as sw
    from topaz_amd.denoising.models import _kind_from_class, load_model
    from topaz_amd.model.unpickle import _PickleModule, _walk

    fake = types.ModuleType('topaz.denoising.models')
    pkgs = {n: types.ModuleType(n) for n in ('topaz', 'topaz.denoising')}

    def make(name):
        cls = type(name, (torch.nn.Module,), {'__module__': 'topaz.denoising.models'})
        setattr(fake, name, cls)
        return cls

    sd = sw.unet_sd(3, nf=8, base_width=7, top_width=3)
    saved = {}
    try:
        sys.modules.update(pkgs)
        sys.modules['topaz.denoising.models'] = fake
        for name in ('UDenoiseNet', 'UDenoiseNet3', 'DenoiseNet', 'UDenoiseNetX'):
            m = make(name)()
            for k, v in sd.items():
                blk, idx, leaf = k.split('.')
                if not hasattr(m, blk):
                    setattr(m, blk, torch.nn.Sequential())
                seq = getattr(m, blk)
                while len(seq) <= int(idx):
                    seq.append(torch.nn.Identity())
                if not isinstance(seq[int(idx)], torch.nn.Conv2d):
                    co, ci, kk = sd[f'{blk}.{idx}.weight'].shape[:3]
                    seq[int(idx)] = torch.nn.Conv2d(ci, co, kk)
                getattr(seq[int(idx)], leaf).data = torch.from_numpy(v)
            saved[name] = str(tmp_path / (name + '.sav'))
            torch.save(m, saved[name])
    finally:
        for n in list(pkgs) + ['topaz.denoising.models']:
            sys.modules.pop(n, None)
    obj = torch.load(saved['UDenoiseNet'], map_location='cpu', weights_only=False, pickle_module=_PickleModule)
    got = {}
    _walk(obj, '', got)
    assert _kind_from_class(obj, got, 'x') == 'unet' and set(got) == set(sd)
    obj3 = torch.load(saved['UDenoiseNet3'], map_location='cpu', weights_only=False, pickle_module=_PickleModule)
    assert _kind_from_class(obj3, got, 'x') == 'unet3'
    with pytest.raises(NotImplementedError, match='fcnet'):
        load_model(saved['DenoiseNet'])
    with pytest.raises(NotImplementedError, match='UDenoiseNetX'):
        load_model(saved['UDenoiseNetX'])


def test_particle_pixel_count_keeps_the_reference_cube_quirk():
    """topaz/stats.py:17-26 builds its mask on a CUBE of side 2r + 1 whatever `dims` is: in 2-D the disc is counted once per
    z plane.  Values below computed with the reference's own formula (np.meshgrid over three axes)."""
    from topaz_amd.stats import calculate_pi, pixels_given_radius
    want = {(1, 2): 15, (1, 3): 7, (3, 2): 203, (3, 3): 123, (7, 2): 2235, (7, 3): 1419, (14, 2): 17777, (14, 3): 11513}
    for (r, d), n in want.items():
        assert pixels_given_radius(r, dims=d) == n, (r, d)
    assert abs(calculate_pi(300, 14, 4096 * 4096) - 17777 * 300 / 4096 ** 2) < 1e-12


def test_pick_rows_are_the_reference_f_string_byte_for_byte():
    """tpz_format_picks (host code of the library, csrc/host_io.hip) writes the rows of `topaz extract`'s pick table
    (topaz/extract.py:341-354: f'{name}\\t{x}\\t{y}\\t{score}' with numpy scalars -- a float32 score prints with the digits of
    its float64 value, CPython's repr: shortest round-trip digits, fixed notation for 1e-4 <= |v| < 1e16, '.0' on integers,
    two-digit exponents) -- random scores, every notation boundary, signed zero, subnormals, inf / nan; 2-D and 3-D rows,
    int64 coordinates as np.round(coords * scale).astype(int) hands them over."""
    from topaz_amd.utils.files import format_pick_rows
    rs = np.random.RandomState(3)
    n = 20000
    coords = rs.randint(0, 11520, size=(n, 2)).astype(np.int32)
    scores = (rs.randn(n) * 6 - 3).astype(np.float32)
    special = np.array([0.0, -0.0, 1.0, -1.0, 1e-4, 9.9e-5, 1e-5, 123456.0, 1e15, 1e16, 1.5e16, 1e17, 3.4e38, 1e-38, 1e-45,
                        np.inf, -np.inf, np.nan, 0.1, 100.0, 16777216.0, 0.5, 2.5e-5, -7e-10, 1e22, 9999999.0, 0.001, -0.06772084],
                       dtype=np.float32)
    scores[:len(special)] = special
    scores[len(special):2 * len(special)] = -special
    ref = ''.join('\t'.join(['mic_a'] + [format(v, '') for v in row] + [format(s, '')]) + '\n' for row, s in zip(coords, scores))
    assert format_pick_rows('mic_a', coords, scores, 2) == ref.encode()
    c3 = rs.randint(0, 512, size=(300, 3)).astype(np.int64)
    ref3 = ''.join(f'tomo\t{r[0]}\t{r[1]}\t{r[2]}\t{s}\n' for r, s in zip(c3, scores[:300]))
    assert format_pick_rows('tomo', c3, scores[:300], 3) == ref3.encode()
    assert format_pick_rows('x', np.zeros((0, 2), dtype=np.int32), np.zeros(0, dtype=np.float32)) == b''
