"""state_dict -> layer program (the manifest tpz_model_load consumes).

Host-side mirror of the module graphs of the reference: the filled ResNets
(topaz/model/features/resnet.py:87-92,153-164,185-202,227-251,280-339), the filled
BasicConv stacks conv127/63/31 (topaz/model/features/basic.py:12-111, model/factory.py:15-25),
the 1x1 LinearClassifier head (topaz/model/classifier.py:29,64-66) and the denoisers
(topaz/denoising/models.py:52-175,178-244,452-564; topaz/filters.py:40-48).

Eval-mode BatchNorm directly after a convolution is folded into that convolution's weights and
bias; ResidA's bn1 sits after the residual add (resnet.py:199-201) and is kept as a
per-channel affine in the conv epilogue.  The classifier head is fused into the last feature
convolution, so the 4u-channel feature map never reaches HBM.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np

from ..runtime import LayerProgram

BN_EPS = 1e-5


def _np(sd) -> Dict[str, np.ndarray]:
    out = {}
    for k, v in sd.items():
        if hasattr(v, 'detach'):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    return out


def _bn_affine(sd, prefix) -> Tuple[np.ndarray, np.ndarray]:
    """eval-mode BatchNorm as y = x*scale + shift"""
    g, b = sd[prefix + '.weight'].astype(np.float64), sd[prefix + '.bias'].astype(np.float64)
    m, v = sd[prefix + '.running_mean'].astype(np.float64), sd[prefix + '.running_var'].astype(np.float64)
    scale = g / np.sqrt(v + BN_EPS)
    return scale, b - m * scale


def _fold_bn(w, b, sd, bn_prefix):
    """conv -> BN  ==  conv with w*scale, (b*scale + shift)"""
    if bn_prefix + '.running_mean' not in sd:
        return w.astype(np.float32), None if b is None else b.astype(np.float32)
    scale, shift = _bn_affine(sd, bn_prefix)
    w2 = w.astype(np.float64) * scale.reshape((-1,) + (1,) * (w.ndim - 1))
    b2 = shift if b is None else b.astype(np.float64) * scale + shift
    return w2.astype(np.float32), b2.astype(np.float32)


# ---- ResNet6 / ResNet8 / ResNet16 -------------------------------------------------------------------
def resnet_modules(arch: str, pooling: bool = False) -> List[dict]:
    """module list of make_modules (resnet.py:254-339).  pooling=False: the convolutions stride where the reference strides
    (`stride = 2 if pooling is None else 1`); pooling=True (`topaz train --pooling max`): stride-1 convolutions and a
    MaxPool(3, stride 2) after the strided positions.  ResNet6 always pools."""
    B = lambda k, s=1: dict(kind='basic', k=k, stride=s, og_dil=1)
    R = lambda d=1, s=1: dict(kind='resid', d=d, stride=s)
    P = lambda: dict(kind='pool', k=3, stride=2)
    s = 1 if pooling else 2
    pool = (lambda: [P()]) if pooling else (lambda: [])         # a fresh dict per position: resnet_fill annotates them
    if arch == 'resnet6':       # resnet.py:254-277
        return [B(5), P(), R(4), P(), R(2), B(5)]
    if arch == 'resnet8':       # resnet.py:280-306
        return [B(7, s)] + pool() + [R(2), R(2, s)] + pool() + [R(2), B(5)]
    if arch == 'resnet16':      # resnet.py:309-339
        return [B(7), R(1, s)] + pool() + [R(), R(), R(), R(1, s)] + pool() + [R(), R(), B(5)]
    raise ValueError(f'unknown ResNet architecture {arch!r}')


def resnet_width(mods: Sequence[dict]) -> int:
    """insize_from_outsize(modules, 1) (model/utils.py:39-68): ResidA reports kernel 2*d+3, dilation 1; MaxPool kernel 3"""
    out = 1
    for m in reversed(mods):
        k = m['k'] if m['kind'] in ('basic', 'pool') else 2 * m['d'] + 3
        out = (out - 1) * m['stride'] + 1 + (k - 1)
    return out


def resnet_fill(mods: Sequence[dict]) -> int:
    """ResNet.fill (resnet.py:227-232): strides become dilations of everything downstream"""
    stride = 1
    for m in mods:
        if m['kind'] == 'basic':
            m['dil'] = m['og_dil'] * stride            # BasicConv.fill, resnet.py:87-92
        elif m['kind'] == 'pool':
            m['dil'] = stride                          # MaxPool.fill, resnet.py:30-36
        else:
            m['dil0'], m['dil1'] = stride, m['d'] * stride   # ResidA.fill, resnet.py:153-164
        stride *= m['stride']
    return stride


def pack_resnet(arch: str, sd, dims: int = 2, pooling: bool = False) -> Tuple[LayerProgram, int]:
    """dims = 3: the same graph over Conv3d / BatchNorm3d weights (resnet.py:56-63,115-123; `--dims 3`)"""
    sd = _np(sd)
    mods = resnet_modules(arch, pooling)
    width = resnet_width(mods)
    resnet_fill(mods)
    if sd['features.features.0.conv.weight'].ndim != dims + 2:
        raise ValueError(f'{arch}: the weights are {sd["features.features.0.conv.weight"].ndim - 2}-D, dims = {dims} was asked for')
    P = LayerProgram(dims)
    cur = 0
    pre0 = 'features.features.'
    head_w = sd['classifier.weight'].reshape(-1)
    head_b = float(sd['classifier.bias'].reshape(-1)[0])
    for i, m in enumerate(mods):
        pre = f'{pre0}{i}.'
        last = i == len(mods) - 1
        pad = width // 2 if i == 0 else 0            # F.pad(x, width//2) once, then valid convs (resnet.py:246-249)
        if m['kind'] == 'pool':
            cur = P.maxpool(cur, m['k'], m['dil'])     # parameter-free, but it owns index i of features.features
        elif m['kind'] == 'basic':
            w, b = _fold_bn(sd[pre + 'conv.weight'], sd.get(pre + 'conv.bias'), sd, pre + 'bn')
            kw = dict(head_w=head_w, head_b=head_b) if last else {}
            cur = P.conv(cur, w, b, dil=m['dil'], pad=pad, slope=0.0, **kw)
        else:
            w0, b0 = _fold_bn(sd[pre + 'conv0.weight'], sd.get(pre + 'conv0.bias'), sd, pre + 'bn0')
            t = P.conv(cur, w0, b0, dil=m['dil0'], pad=pad, slope=0.0)
            res = cur
            if pre + 'proj.weight' in sd:
                res = P.conv(cur, sd[pre + 'proj.weight'], None, dil=1, pad=0, slope=1.0)
            ps = pt = None
            if pre + 'bn1.running_mean' in sd:
                ps, pt = _bn_affine(sd, pre + 'bn1')
            cur = P.conv(t, sd[pre + 'conv1.weight'], sd.get(pre + 'conv1.bias'), dil=m['dil1'], pad=0, slope=0.0,
                         res=res, res_crop=m['dil0'] + m['dil1'], post_scale=ps, post_shift=pt)
    if mods[-1]['kind'] != 'basic':
        raise ValueError('head fusion expects the feature stack to end with a BasicConv')
    return P, width


# ---- conv127 / conv63 / conv31 ---------------------------------------------------------------------
BASIC_SIZES = {'conv127': (7, 5, 5, 5, 5), 'conv63': (7, 5, 5, 5), 'conv31': (7, 5, 5)}


def basic_width(sizes: Sequence[int]) -> int:
    strides = [2] * (len(sizes) - 1) + [1]
    out = 1
    for k, s in zip(reversed(sizes), reversed(strides)):
        out = (out - 1) * s + 1 + (k - 1)
    return out


def basic_fill_dilations(n_convs: int, has_bn: bool, dropout: bool = False) -> List[int]:
    """dilation BasicConv.fill() gives each conv (basic.py:81-89).  fill() walks the layer list zipped with `strides`,
    and the constructor appends no `strides` entry for its nn.Dropout layers (basic.py:57-58,69-70): without dropout the
    result is the cumulative stride 1, 2, 4, ...; a model trained with --dropout has the pairs slip by one per Dropout
    (conv31: 1, 4, 4) and that is what upstream's `extract` then scores with -- reproduced here, not corrected."""
    kinds, strides = [], []
    for i in range(n_convs):
        last = i == n_convs - 1
        kinds.append('conv')
        strides.append(1 if last else 2)
        if has_bn:
            kinds.append('bn')
            strides.append(1)
        kinds.append('act')
        strides.append(1)
        if dropout:
            kinds.append('drop')
    dils, stride = [], 1
    for kind, st in zip(kinds, strides):
        if kind == 'conv':
            dils.append(stride)
        stride *= st
    return dils + [1] * (n_convs - len(dils))        # convs past the end of the zip keep dilation 1 (and their stride 1)


def pack_basicconv(sizes: Sequence[int], sd, dropout: bool = False, dims: int = 2) -> Tuple[LayerProgram, int]:
    """filled basic.BasicConv (basic.py:81-89: dilation = cumulative stride 1,2,4,..) + 1x1 head.  `sd` is numbered
    without Dropout modules (unpickle.py renumbers); `dropout` only selects upstream's fill pattern for such models.
    dims = 3: the same stack over Conv3d / BatchNorm3d weights (basic.py:23-27)."""
    sd = _np(sd)
    if sd['features.features.0.weight'].ndim != dims + 2:
        raise ValueError(f'BasicConv stack: the weights are {sd["features.features.0.weight"].ndim - 2}-D, dims = {dims} was asked for')
    has_bn = any(k.endswith('running_mean') for k in sd)
    width = basic_width(sizes)
    dils = basic_fill_dilations(len(sizes), has_bn, dropout)
    P = LayerProgram(dims)
    pre = 'features.features.'
    head_w = sd['classifier.weight'].reshape(-1)
    head_b = float(sd['classifier.bias'].reshape(-1)[0])
    cur, idx = 0, 0
    for li, k in enumerate(sizes):
        w, b = sd[f'{pre}{idx}.weight'], sd.get(f'{pre}{idx}.bias')
        idx += 1
        if has_bn:
            w, b = _fold_bn(w, b, sd, f'{pre}{idx}')
            idx += 1
        slope = float(np.asarray(sd[f'{pre}{idx}.weight']).reshape(-1)[0])    # nn.PReLU(): one shared slope
        idx += 1
        last = li == len(sizes) - 1
        kw = dict(head_w=head_w, head_b=head_b) if last else {}
        cur = P.conv(cur, w, b, dil=dils[li], pad=width // 2 if li == 0 else 0, slope=slope, **kw)
    return P, width


# ---- denoisers -------------------------------------------------------------------------------------
def pack_unet(sd, depth: int, dims: int = 2, noise_only: bool = False, no_skip=()) -> LayerProgram:
    """UDenoiseNet (depth 5), UDenoiseNetSmall (depth 3), UDenoiseNet3D (depth 5, dims 3), UDenoiseNet3 (noise_only) and
    UDenoiseNet2 (no_skip = (2, 1)).
    Upsample + concat never materialise: the consumer conv reads two sources (src nearest-upsampled).
    noise_only (UDenoiseNet3, denoising/models.py:447: `y = x - self.dec1(h)`): the last conv runs with negated weights and
    bias and adds the input as a residual of its own size.
    no_skip (UDenoiseNet2, models.py:321-338): decoder levels whose first conv reads the upsampled tensor ALONE.  The layer
    program has no upsample-without-concat; the skip tensor that fixes the upsampled size is concatenated all the same, with
    zero weights on its channels -- the same sums, a few dead multiply-adds on a user-trained architecture."""
    sd = _np(sd)
    P = LayerProgram(dims)

    def c(src, name, slope=0.1, src2=-1, res=-1, negate=False, dead_channels=0):
        w = sd[name + '.weight']
        b = sd.get(name + '.bias')
        if dead_channels:
            w = np.concatenate([w, np.zeros((w.shape[0], dead_channels) + w.shape[2:], dtype=w.dtype)], axis=1)
        if negate:
            w, b = -w, (None if b is None else -b)
        return P.conv(src, w, b, dil=1, pad=w.shape[-1] // 2, slope=slope, src2=src2, res=res)

    skips, skip_ch = [0], [1]
    h = 0
    for i in range(1, depth + 1):
        h = P.maxpool2(c(h, f'enc{i}.0'))
        skips.append(h)
        skip_ch.append(sd[f'enc{i}.0.weight'].shape[0])
    h = c(h, f'enc{depth + 1}.0')
    for lvl in range(depth, 0, -1):
        dead = skip_ch[lvl - 1] if lvl in no_skip else 0
        h = c(h, f'dec{lvl}.0', src2=skips[lvl - 1], dead_channels=dead)
        h = c(h, f'dec{lvl}.2')
    h = c(h, 'dec1.4', slope=1.0, res=0 if noise_only else -1, negate=noise_only)
    return P


def pack_fcnn(sd) -> LayerProgram:
    sd = _np(sd)
    P = LayerProgram(2)
    h = 0
    for name, slope in (('net.0', 0.1), ('net.2', 0.1), ('net.4', 1.0)):
        w = sd[name + '.weight']
        h = P.conv(h, w, sd.get(name + '.bias'), pad=w.shape[-1] // 2, slope=slope)
    return P


def pack_filter(weight, bias=None, dims: int = 2) -> LayerProgram:
    """AffineDenoise / GaussianDenoise: one 1->1 'same' convolution (filters.py:40-80)"""
    w = np.asarray(weight, dtype=np.float32)
    w = w.reshape((1, 1) + w.shape[-dims:])
    P = LayerProgram(dims)
    P.conv(0, w, None if bias is None else np.asarray(bias, dtype=np.float32).reshape(1), pad=w.shape[-1] // 2,
           slope=1.0)
    return P
