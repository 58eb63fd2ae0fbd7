"""The benchmarked path at the benchmark's own size and on the benchmark's own networks, ABSOLUTE 1e-4.

bench.py runs denoise (`unet` architecture: base 11 / top 5, 48 filters, seed 11, -s 1024 -p 500) -> score
(`resnet8` architecture, 64 units, seed 7, filled, head fused) -> NMS (r = 14, t = -6) on 4096 x 4096 micrographs
with the default (2xf16) convolution kernels.  These tests pin exactly that configuration against the oracle:
  * size-independent properties at 4096^2 (translation equivariance of the filled net: windows of the score map vs
    the oracle on the window's own crop; independence of the denoise patches: a patch centre vs the oracle's
    `_denoise` of that patch crop) -- on the default path and, for the scoring net, also the exact-fp32 path;
  * the chained workload of BASELINE config 4 (denoise -> score(denoised) -> NMS) against the oracle chain, with
    the pretrained nets SURVEY 8(d) names for C4 and with the bench's nets;
  * small-magnitude activations (the lo halves of the split format have an absolute floor of 2^-25).
Every tolerance in this file is an absolute |a - b| <= 1e-4 (BASELINE.json north_star), never scaled."""
import numpy as np
import pytest
import torch

from oracle import denoising as oden
from oracle import nms as onms
from oracle import scoring as oscoring

pytestmark = pytest.mark.gpu
ATOL = 1e-4


def _abs(a, b):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max())


def _bench_resnet():
    """the scoring net of bench.py: same generator, same seed; the head is calibrated here with the ORACLE's logits
    (tests/test_cpu_host.py pins that bench.py's recorded calibration constants give bit-identical weights)"""
    from topaz_amd.model.classifier import LinearClassifier
    sd = oscoring.synthetic_resnet_sd('resnet8', 64, 7)
    m = LinearClassifier('resnet8', sd)
    m.eval(); m.fill(); m.cuda()
    return m, sd


def _bench_unet():
    from topaz_amd.denoise import Denoise
    from topaz_amd.denoising.models import DenoiseNet
    sd = oden.synthetic_unet_sd(11, nf=48, base_width=11, top_width=5)
    return Denoise(DenoiseNet('unet', sd)), sd


WINDOWS = ((0, 0), (1900, 2100), (3840, 0), (3840, 3840), (777, 3840), (2048, 1000))


def _check_windows(y, x, sd, arch, p, size=256):
    H, W = x.shape
    worst = 0.0
    for (y0, x0) in WINDOWS:
        ys, xs = max(0, y0 - p), max(0, x0 - p)
        ye, xe = min(H, y0 + size + p), min(W, x0 + size + p)
        ref = oscoring.score(arch, sd, x[ys:ye, xs:xe])
        ref_win = ref[y0 - ys:y0 - ys + size, x0 - xs:x0 - xs + size]
        e = _abs(y[y0:y0 + size, x0:x0 + size], ref_win)
        assert e <= ATOL, (y0, x0, e)
        worst = max(worst, e)
    return worst


def test_bench_resnet8_u64_4096_windows_vs_oracle(gpu_ctx):
    """resnet8-u64 (seed 7) on a 4096^2 micrograph, default 2xf16 path and exact-fp32 path: every window equals the
    oracle on the window's own crop (71-pixel receptive field -> 35 halo) to 1e-4 absolute; logits span about
    [-25, +8] here."""
    m, sd = _bench_resnet()
    x = np.random.RandomState(1000).randn(4096, 4096).astype(np.float32)
    xt = torch.from_numpy(x).cuda()[None, None]
    dm = m.device_model
    before = dm.split_stats()
    assert before[0], 'the bench net must be eligible for the 2xf16 path'
    y = m(xt)[0, 0].cpu().numpy()
    after = dm.split_stats()
    assert after[1] == before[1] + 1 and after[2] == before[2], 'expected one 2xf16 forward without an fp32 re-run'
    assert y.min() < -15 and y.max() > 0
    e_split = _check_windows(y, x, sd, 'resnet8', 35)
    gpu_ctx.set_exact(True)
    try:
        y32 = m(xt)[0, 0].cpu().numpy()
    finally:
        gpu_ctx.set_exact(False)
    e_f32 = _check_windows(y32, x, sd, 'resnet8', 35)
    # the whole map: 2xf16 vs the fp32 kernels (16.7 M logits), absolute
    assert _abs(y, y32) <= ATOL
    print(f'resnet8-u64 4096^2: max |split - oracle| {e_split:.2e}, |fp32 - oracle| {e_f32:.2e}, |split - fp32| {_abs(y, y32):.2e}')


def test_cli_default_resnet16_u64_4096_windows_vs_oracle_and_float64(gpu_ctx):
    """`topaz extract` defaults to -m resnet16 (= 64 units; commands/extract.py:16-53, factory.py:36-51): the deepest network of
    the path (16 convolutions, K up to 3 200 per sum) and the one with the least headroom under the 1e-4 bar (6.3e-5 against
    float64 on a 512^2 image, torch fp32 itself 3.6e-5).  At 4096^2 (seeded weights, head calibrated like the pretrained nets:
    logits about [-25, +8]): windows of the score map -- corners, edges, interior -- against the oracle's fp32 AND float64
    evaluation of the window's own crop: <= 1e-4 absolute against both, and against float64 no more than 2x what torch's own
    fp32 evaluation is off by; no fp32 re-run."""
    from topaz_amd.model.classifier import LinearClassifier
    sd = oscoring.synthetic_resnet_sd('resnet16', 64, 7)
    m = LinearClassifier('resnet16', sd)
    m.eval(); m.fill(); m.cuda()
    halo = m.width // 2
    x = np.random.RandomState(1001).randn(4096, 4096).astype(np.float32)
    dm = m.device_model
    before = dm.split_stats()
    y = m(torch.from_numpy(x).cuda()[None, None])[0, 0].cpu().numpy()
    after = dm.split_stats()
    assert after[1] == before[1] + 1 and after[2] == before[2], 'expected one 2xf16 forward without an fp32 re-run'
    assert y.shape == x.shape and y.min() < -12 and y.max() > -2
    size, worst, worst64, worst_t = 192, 0.0, 0.0, 0.0
    for (y0, x0) in ((0, 0), (3904, 3904), (1900, 2100), (0, 2048), (2048, 3904)):
        ys, xs = max(0, y0 - halo), max(0, x0 - halo)
        ye, xe = min(4096, y0 + size + halo), min(4096, x0 + size + halo)
        crop = x[ys:ye, xs:xe]
        win = lambda a: a[y0 - ys:y0 - ys + size, x0 - xs:x0 - xs + size]
        ref32, ref64 = win(oscoring.score('resnet16', sd, crop)), win(oscoring.score('resnet16', sd, crop, dtype=torch.float64))
        got = y[y0:y0 + size, x0:x0 + size]
        e32, e64, et = _abs(got, ref32), _abs(got, ref64), _abs(ref32, ref64)
        assert e32 <= ATOL and e64 <= ATOL, (y0, x0, e32, e64)
        worst, worst64, worst_t = max(worst, e32), max(worst64, e64), max(worst_t, et)
    print(f'resnet16-u64 4096^2 (halo {halo}): max |2xf16 - oracle fp32| {worst:.2e}, |2xf16 - float64| {worst64:.2e}, '
          f'|torch fp32 - float64| {worst_t:.2e}')
    assert worst64 <= 2.0 * worst_t, (worst64, worst_t)


def test_bench_unet_nf48_4096_default_patching_vs_oracle_patches(gpu_ctx):
    """unet b11/t5 nf48 (seed 11) with the CLI-default patching on a 4096^2 N(0,1) micrograph (exactly bench.py's
    denoise stage): the centre of a patch equals the oracle's `_denoise` of that patch crop alone (denoise.py:307-322)
    to 1e-4 absolute -- a corner patch (1524^2 crop) and an interior one (2024^2 crop)."""
    d, sd = _bench_unet()
    x = np.random.RandomState(1000).randn(4096, 4096).astype(np.float32)
    dm = d.model.device_model
    before = dm.split_stats()
    y = d.denoise_device(torch.from_numpy(x).cuda(), 1024, 500).cpu().numpy()
    after = dm.split_stats()
    assert after[1] > before[1] and after[2] == before[2], 'expected 2xf16 forwards without an fp32 re-run'
    tsd = oden.to_torch_sd(sd)
    for (i, j) in ((0, 3072), (2048, 1024)):
        si, ei, sj, ej = max(0, i - 500), min(4096, i + 1524), max(0, j - 500), min(4096, j + 1524)
        ref = oden.denoise_whole('unet', tsd, torch.from_numpy(x[si:ei, sj:ej].copy()))
        ref = ref[i - si:i - si + 1024, j - sj:j - sj + 1024]
        e = _abs(y[i:i + 1024, j:j + 1024], ref)
        assert e <= ATOL, (i, j, e)
    # the same micrograph with every tensor of every patch computed in full (patch windows off): not one bit differs
    try:
        gpu_ctx.set_roi(False)
        y_full = d.denoise_device(torch.from_numpy(x).cuda(), 1024, 500).cpu().numpy()
    finally:
        gpu_ctx.set_roi(True)
    assert np.array_equal(y, y_full)


def _picks_equivalent(s, c, so, co, ref_map, r, tol, thr=-6.0):
    """pick tables identical, or every differing pick explained by a competitor within `tol` of it in the oracle's map"""
    got, want = set(map(tuple, np.asarray(c).tolist())), set(map(tuple, np.asarray(co).tolist()))
    H, W = ref_map.shape
    for (px, py) in got ^ want:
        win = ref_map[max(0, py - r):py + r + 1, max(0, px - r):px + r + 1]
        near_thr = abs(ref_map[py, px] - thr) < tol
        assert near_thr or np.sort(win.ravel())[-1] - ref_map[py, px] < tol, (px, py)
    return len(got ^ want), len(want)


@pytest.mark.parametrize('nets', ['pretrained', 'bench'])
def test_chained_config4_denoise_score_nms_vs_oracle_chain(gpu_ctx, nets):
    """BASELINE config 4 on one GPU: denoise -> score(denoised) -> NMS, all on the device (what bench.py times and
    what `topaz denoise` + `topaz extract` compose to), against the oracle chain on the same micrograph.
    'pretrained': unet-v0.2.1 + resnet8_u32 (the nets SURVEY 8(d) names for C4) at 1536^2 with -s 1024 -p 500;
    'bench': the seeded unet nf48 + resnet8-u64 of bench.py at 1100^2 with -s 512 -p 250."""
    from topaz_amd.algorithms import non_maximum_suppression
    from topaz_amd.denoise import Denoise
    from topaz_amd.model.factory import load_model
    if nets == 'pretrained':
        size, patch, pad, r = 1536, 1024, 500, 8
        d = Denoise('unet-v0.2.1')
        sd_d = {k: v.numpy() for k, v in d.model.state_dict().items()}
        m = load_model('resnet8_u32')
        m.eval(); m.fill(); m.cuda()
        sd_s = {k: v.numpy() for k, v in m.state_dict().items()}
    else:
        size, patch, pad, r = 1100, 512, 250, 14
        d, sd_d = _bench_unet()
        m, sd_s = _bench_resnet()
    rs = np.random.RandomState(1000)
    x = rs.randn(size, size).astype(np.float32)
    # dark blobs ("particles") under the noise, so that the pretrained detector has something to find after denoising
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32)
    for cy, cx in rs.randint(40, size - 40, size=(120, 2)):
        x -= 2.5 * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * 9.0 ** 2)).astype(np.float32)
    # oracle chain
    den_ref = oden.denoise('unet', sd_d, x, patch, pad)
    log_ref = oscoring.score('resnet8', sd_s, den_ref)
    # threshold: the CLI default -6 where the map has enough candidates above it, else the map's own 90 % quantile
    thr = -6.0 if (log_ref > -6.0).mean() > 0.02 else float(np.float32(np.quantile(log_ref, 0.9)))
    so, co = onms.nms2d(log_ref, r, thr)
    assert len(so) > 50, (len(so), thr)
    # device chain
    den = d.denoise_device(torch.from_numpy(x).cuda(), patch, pad)
    logits = m(den[None, None])[0, 0]
    s, c = non_maximum_suppression(logits, r, threshold=thr)
    e_den, e_log = _abs(den, den_ref), _abs(logits, log_ref)
    # stage parity on identical inputs (score of the DEVICE's denoised image vs the oracle scoring that same image) ...
    e_stage = _abs(logits, oscoring.score('resnet8', sd_s, den.cpu().numpy()))
    print(f'chain[{nets}] {size}^2: |den| {e_den:.2e}  |logit| chain {e_log:.2e} stage {e_stage:.2e}  picks {len(so)} thr {thr:.3f}')
    assert e_den <= ATOL and e_stage <= ATOL
    # ... and the chain end to end: the 1e-4 of the first stage passes through the second
    assert e_log <= ATOL
    n_diff, n = _picks_equivalent(s, c, so, co, log_ref, r, 2 * ATOL, thr)
    print(f'{nets}: {n} picks, {n_diff} differ from the oracle chain (each explained by a competitor within {2 * ATOL:g} of it)')
    assert n_diff <= max(2, n // 100), (n_diff, n)
    # NMS of the oracle's own map on the device is bit-exact
    s2, c2 = non_maximum_suppression(log_ref, r, threshold=thr)
    assert np.array_equal(c2, co) and np.array_equal(s2, so)


def _sample_windows(n, size, H, W, seed):
    """n window origins of size^2: the four corners, the four edge midpoints, the rest uniformly at random"""
    fixed = [(0, 0), (0, W - size), (H - size, 0), (H - size, W - size), (0, W // 2), (H - size, W // 2), (H // 2, 0),
             (H // 2, W - size)]
    rs = np.random.RandomState(seed)
    rest = [(int(rs.randint(0, H - size + 1)), int(rs.randint(0, W - size + 1))) for _ in range(n - len(fixed))]
    return fixed + rest


def test_config4_chain_at_4096_every_patch_and_4096_sampled_logits(gpu_ctx):
    """BASELINE config 4 at its own size, on the bench's own networks, stage by stage and end to end, ABSOLUTE 1e-4:
      * denoise (-s 1024 -p 500): EVERY one of the 16 patch centres against the oracle's `_denoise` of that patch's crop
        (denoise.py:307-322) -- the oracle's denoised micrograph is assembled from them;
      * score: 4096 logits sampled as 64 windows of 8 x 8 (corners, edges, random) + six 256^2 windows, (a) the device's
        scorer against the oracle scoring the DEVICE's denoised image (stage parity on identical input), (b) the device
        chain against the oracle chain (oracle scorer on the oracle's denoised image);
      * NMS (r = 14, t = -6): the device's pick table equals the C oracle's greedy suppression of the device's score map,
        bit for bit, all 16.7 M pixels."""
    from topaz_amd.algorithms import non_maximum_suppression
    d, sd_d = _bench_unet()
    m, sd_s = _bench_resnet()
    H = W = 4096
    x = np.random.RandomState(1000).randn(H, W).astype(np.float32)
    den_t = d.denoise_device(torch.from_numpy(x).cuda(), 1024, 500)
    logits_t = m(den_t[None, None])[0, 0]
    s, c = non_maximum_suppression(logits_t, 14, threshold=-6.0)
    den, logits = den_t.cpu().numpy(), logits_t.cpu().numpy()
    # ---- denoise: all 16 patches
    tsd = oden.to_torch_sd(sd_d)
    den_ref = np.empty_like(den)
    worst_den = 0.0
    for i in range(0, H, 1024):
        for j in range(0, W, 1024):
            si, ei, sj, ej = max(0, i - 500), min(H, i + 1524), max(0, j - 500), min(W, j + 1524)
            ref = oden.denoise_whole('unet', tsd, torch.from_numpy(x[si:ei, sj:ej].copy()))
            den_ref[i:i + 1024, j:j + 1024] = ref[i - si:i - si + 1024, j - sj:j - sj + 1024]
            e = _abs(den[i:i + 1024, j:j + 1024], den_ref[i:i + 1024, j:j + 1024])
            assert e <= ATOL, (i, j, e)
            worst_den = max(worst_den, e)
    # ---- score: sampled windows, stage-wise and end to end
    p = 35
    wins = [(y0, x0, 8) for (y0, x0) in _sample_windows(64, 8, H, W, 5)] + [(y0, x0, 256) for (y0, x0) in WINDOWS]
    worst_stage = worst_chain = 0.0
    for (y0, x0, size) in wins:
        ys, xs, ye, xe = max(0, y0 - p), max(0, x0 - p), min(H, y0 + size + p), min(W, x0 + size + p)
        got = logits[y0:y0 + size, x0:x0 + size]
        for src, name in ((den, 'stage'), (den_ref, 'chain')):
            ref = oscoring.score('resnet8', sd_s, src[ys:ye, xs:xe])[y0 - ys:y0 - ys + size, x0 - xs:x0 - xs + size]
            e = _abs(got, ref)
            assert e <= ATOL, (name, y0, x0, size, e)
            if name == 'stage':
                worst_stage = max(worst_stage, e)
            else:
                worst_chain = max(worst_chain, e)
    # ---- NMS: the whole table against the C oracle on the same map
    so, co = onms.nms2d(logits, 14, -6.0)
    assert len(so) > 10000
    assert np.array_equal(np.asarray(c), co) and np.array_equal(np.asarray(s), so)
    print(f'config 4 at 4096^2: |den| {worst_den:.2e} over 16 patches; |logit| stage {worst_stage:.2e}, chain {worst_chain:.2e} '
          f'over {sum(w[2] ** 2 for w in wins)} sampled logits; {len(so)} picks identical')


def test_small_magnitude_activations_split_floor(gpu_ctx):
    """The lo half of a split value is an f16 with an absolute floor of 2^-25 (below it the value carries fewer than
    22 bits).  Layer level: activations of magnitude 1e-3 ... 1e-7 through one 2xf16 convolution against float64 --
    the absolute error stays below 2^-25 * sum|w| (each activation is off by at most 2^-25), i.e. far inside 1e-4.
    Network level: the bench's scoring net on a micrograph scaled by 1e-3 (feature maps dominated by the biases,
    tiny conv contributions) still matches the oracle to 1e-4 absolute."""
    import torch.nn.functional as F
    from topaz_amd import runtime as rt
    g = torch.Generator().manual_seed(41)
    for mag in (1e-3, 1e-5, 1e-7):
        x = torch.randn(64, 50, 61, generator=g) * mag
        w = torch.randn(64, 64, 3, 3, generator=g) / 24
        b = torch.randn(64, generator=g) * mag
        ref = F.conv2d(x[None].double(), w.double(), b.double(), dilation=2)[0]
        y, ovf = rt.conv_split(x, w.numpy(), b.numpy(), dil=2, slope=1.0)
        assert not ovf
        mass = F.conv2d(x[None].double().abs(), w.double().abs(), b.double().abs(), dilation=2)[0]   # sum |w||x|
        bound = 2.0 ** -25 * float(w.abs().sum(dim=(1, 2, 3)).max()) + 2.0 ** -21 * float(mass.max()) + 1e-12
        e = _abs(y.double(), ref)
        assert e <= bound, (mag, e, bound)
        assert e <= 1e-6
    m, sd = _bench_resnet()
    x = (np.random.RandomState(6).randn(300, 320) * 1e-3).astype(np.float32)
    y = m(torch.from_numpy(x).cuda()[None, None])[0, 0].cpu().numpy()
    assert _abs(y, oscoring.score('resnet8', sd, x)) <= ATOL
