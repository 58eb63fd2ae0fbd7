"""Oracle (test infrastructure): CPU restatement of the reference's filled scoring networks.

Follows topaz/model/features/resnet.py (BasicConv :50-105, ResidA :108-204, ResNet.fill/forward
:227-251, ResNet8 :280-306, ResNet16 :309-339), topaz/model/features/basic.py (BasicConv :12-111),
topaz/model/classifier.py (LinearClassifier :14-66) and topaz/model/utils.py insize_from_outsize
(:39-68).  The networks are described as plain data (a list of module specs), `fill` performs the
reference's stride -> dilation rewrite on that data, and `forward` evaluates it with torch-CPU
functional ops on a state_dict.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List

import numpy as np
import torch
import torch.nn.functional as F


# ---------------------------------------------------------------------------------------------
# architecture specs (unfilled), mirroring make_modules of the reference
# ---------------------------------------------------------------------------------------------
def _basic(k, stride=1, dilation=1):
    return {'type': 'basic', 'k': k, 'stride': stride, 'og_dilation': dilation, 'dilation': dilation}


def _resid(dilation=1, stride=1):
    # ResidA: conv0 3x3 (dil 1), conv1 3x3 (dil `dilation`, stride `stride`); resnet.py:129-143
    return {'type': 'resid', 'k': 2 * dilation + 3, 'stride': stride, 'conv0_dil': 1, 'conv1_dil': dilation,
            'dilation': 1}


def _pool():
    # MaxPool(3, stride=2) (resnet.py:10-22)
    return {'type': 'pool', 'k': 3, 'stride': 2, 'dilation': 1}


def resnet6_spec(pooling: bool = True) -> List[dict]:
    # resnet.py:263-273: always pooled
    return [_basic(5), _pool(), _resid(dilation=4), _pool(), _resid(dilation=2), _basic(5)]


def _drop():
    # nn.Dropout(p) of `topaz train --dropout p` (resnet.py:296,300,303 / 326,332,336): identity at inference, but it
    # takes an index in features.features
    return {'type': 'drop'}


def resnet8_spec(pooling: bool = False, dropout: bool = False) -> List[dict]:
    # resnet.py:291-303: stride = 2 if pooling is None else 1
    s = 1 if pooling else 2
    p = (lambda: [_pool()]) if pooling else (lambda: [])           # (a fresh dict per position: fill() annotates them)
    d = [_drop()] if dropout else []
    return ([_basic(7, stride=s)] + p() + d + [_resid(dilation=2), _resid(dilation=2, stride=s)] + p() + d +
            [_resid(dilation=2), _basic(5)] + d)


def resnet16_spec(pooling: bool = False, dropout: bool = False) -> List[dict]:
    # resnet.py:320-336
    s = 1 if pooling else 2
    p = (lambda: [_pool()]) if pooling else (lambda: [])
    d = [_drop()] if dropout else []
    return ([_basic(7), _resid(stride=s)] + p() + d + [_resid(), _resid(), _resid(), _resid(stride=s)] + p() + d +
            [_resid(), _resid(), _basic(5)] + d)


def width_of(spec: List[dict]) -> int:
    """receptive field of the unfilled stack: insize_from_outsize(modules, 1) (model/utils.py:39-68).
    The module-level attributes it reads are kernel_size / stride / dilation / padding of BasicConv
    (resnet.py:73-78) and ResidA (resnet.py:138-141: kernel_size = 2*dilation+3, dilation = 1)."""
    out = 1
    for m in spec[::-1]:
        if m['type'] == 'drop':
            continue
        dil = m['dilation'] if m['type'] == 'basic' else 1
        out = (out - 1) * m['stride'] + 1 + (m['k'] - 1) * dil
    return out


def fill(spec: List[dict]) -> int:
    """ResNet.fill (resnet.py:227-232): thread the cumulative stride through the modules, turning
    strides into dilations (BasicConv.fill :87-92, ResidA.fill :153-164).  Returns the total stride."""
    stride = 1
    for m in spec:
        if m['type'] == 'drop':
            continue
        if m['type'] == 'basic':
            m['conv_dil'] = m['og_dilation'] * stride
        elif m['type'] == 'pool':
            m['pool_dil'] = stride                      # MaxPool.fill (resnet.py:30-36)
        else:
            m['conv0_fdil'] = stride
            m['conv1_fdil'] = m['conv1_dil'] * stride
        stride *= m['stride']
    return stride


def _bn(y, sd, prefix):
    if prefix + '.weight' not in sd:
        return y
    return F.batch_norm(y, sd[prefix + '.running_mean'], sd[prefix + '.running_var'], sd[prefix + '.weight'],
                        sd[prefix + '.bias'], training=False, eps=1e-5)


def resnet_forward(x: torch.Tensor, sd: Dict[str, torch.Tensor], spec: List[dict], prefix='features.features.',
                   head=True) -> torch.Tensor:
    """Filled forward of LinearClassifier(ResNetN): x [N,1,(D,)H,W] -> logits [N,1,(D,)H,W].
    resnet.py:243-251 (pad by width//2, Sequential), :101-105 (BasicConv), :185-202 (ResidA),
    classifier.py:64-66 (1x1 head)."""
    dims = x.dim() - 2                         # 2-D micrographs or 3-D tomograms (resnet.py:56-63,115-123,186-197)
    conv = F.conv3d if dims == 3 else F.conv2d
    p = width_of(spec) // 2
    h = F.pad(x, (p, p) * dims)
    for i, m in enumerate(spec):
        pre = f'{prefix}{i}.'
        if m['type'] == 'drop':
            continue
        if m['type'] == 'pool':
            h = (F.max_pool3d if dims == 3 else F.max_pool2d)(h, m['k'], stride=1, dilation=m['pool_dil'])
        elif m['type'] == 'basic':
            h = conv(h, sd[pre + 'conv.weight'], sd.get(pre + 'conv.bias'), dilation=m['conv_dil'])
            h = _bn(h, sd, pre + 'bn')
            h = F.relu(h)
        else:
            d0, d1 = m['conv0_fdil'], m['conv1_fdil']
            t = conv(h, sd[pre + 'conv0.weight'], sd.get(pre + 'conv0.bias'), dilation=d0)
            t = _bn(t, sd, pre + 'bn0')
            t = F.relu(t)
            y = conv(t, sd[pre + 'conv1.weight'], sd.get(pre + 'conv1.bias'), dilation=d1)
            edge = d0 + d1
            xs = h[(slice(None), slice(None)) + (slice(edge, -edge),) * dims]
            if pre + 'proj.weight' in sd:
                xs = conv(xs, sd[pre + 'proj.weight'])
            y = y + xs
            y = _bn(y, sd, pre + 'bn1')      # bn1 comes AFTER the add (resnet.py:199-201)
            h = F.relu(y)
    if head:
        h = conv(h, sd['classifier.weight'], sd['classifier.bias'])
    return h


def basicconv_forward(x: torch.Tensor, sd: Dict[str, torch.Tensor], sizes=(7, 5, 5, 5, 5),
                      prefix='features.features.', head=True, dropout=False) -> torch.Tensor:
    """Filled forward of LinearClassifier(basic.BasicConv(sizes, units)) -- conv127/63/31
    (basic.py:12-111, factory.py:15-25).  Every conv but the last has stride 2 (no pooling), so
    fill() gives dilations 1,2,4,... (basic.py:81-89); pad = width//2 with width from the strided
    stack (basic.py:72,105-109).  Module indices advance conv[,bn],act (basic.py:46-69)."""
    has_bn = any(k.endswith('running_mean') for k in sd)
    strides = [2] * (len(sizes) - 1) + [1]
    width = 1
    for k, s in zip(sizes[::-1], strides[::-1]):
        width = (width - 1) * s + 1 + (k - 1)
    p = width // 2
    dims = sd[f'{prefix}0.weight'].dim() - 2          # Conv2d or Conv3d weights (basic.py:19-27)
    conv = F.conv3d if dims == 3 else F.conv2d
    h = F.pad(x, (p, p) * dims)
    # fill() zips the layers with `strides`, which has no entry for Dropout layers (basic.py:57-58,69-70,81-89): a model
    # built with dropout > 0 gets the dilations this slipped walk gives (conv31: 1,4,4), any other 1,2,4,...
    kinds, zs = [], []
    for st in strides:
        kinds += ['conv'] + (['bn'] if has_bn else []) + ['act'] + (['drop'] if dropout else [])
        zs += [st] + ([1] if has_bn else []) + [1]
    dils, cum = [], 1
    for kind, st in zip(kinds, zs):
        if kind == 'conv':
            dils.append(cum)
        cum *= st
    dils += [1] * (len(sizes) - len(dils))
    idx = 0
    for dil in dils:
        h = conv(h, sd[f'{prefix}{idx}.weight'], sd.get(f'{prefix}{idx}.bias'), dilation=dil)
        idx += 1
        if has_bn:
            h = _bn(h, sd, f'{prefix}{idx}')
            idx += 1
        h = F.prelu(h, sd[f'{prefix}{idx}.weight'])
        idx += 1 + (1 if dropout else 0)              # (Dropout: identity in eval mode)
    if head:
        h = conv(h, sd['classifier.weight'], sd['classifier.bias'])
    return h


ARCH_SPECS = {'resnet6': resnet6_spec, 'resnet8': resnet8_spec, 'resnet16': resnet16_spec}
BASIC_SIZES = {'conv127': (7, 5, 5, 5, 5), 'conv63': (7, 5, 5, 5), 'conv31': (7, 5, 5)}


def to_torch_sd(sd) -> Dict[str, torch.Tensor]:
    return OrderedDict((k, torch.as_tensor(np.asarray(v)) if not torch.is_tensor(v) else v) for k, v in sd.items())


@torch.no_grad()
def score(arch: str, sd, x: np.ndarray, num_threads: int = 0, pooling: bool = False, dropout: bool = False,
          dtype=torch.float32) -> np.ndarray:
    """logits of one [H,W] image (or [D,H,W] tomogram, with 3-D weights) with the filled network `arch` (what
    extract.py:247-249 computes).  dropout=True: `sd` is the state_dict of a model built with dropout > 0 (upstream's
    numbering, Dropout modules included).  dtype=torch.float64 evaluates the same fp32 weights and input in double precision
    (the yardstick of the precision tests: how far is an fp32 evaluation from the exact result?)."""
    if num_threads:
        torch.set_num_threads(num_threads)
    sd = to_torch_sd(sd)
    xt = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))[None, None]
    if dtype != torch.float32:
        sd = OrderedDict((k, v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items())
        xt = xt.to(dtype)
    if arch in ARCH_SPECS:
        spec = ARCH_SPECS[arch](pooling, dropout) if arch != 'resnet6' else resnet6_spec()
        fill(spec)
        y = resnet_forward(xt, sd, spec)
    elif arch in BASIC_SIZES:
        y = basicconv_forward(xt, sd, BASIC_SIZES[arch], dropout=dropout)
    else:
        raise ValueError(arch)
    return y[0, 0].numpy()


# ---------------------------------------------------------------------------------------------
# seeded synthetic weights for architectures whose pretrained blobs are absent: generated by
# tools/synth_weights.py (pure NumPy); here the ResNet head is calibrated with this oracle's own forward pass
# ---------------------------------------------------------------------------------------------
def synthetic_resnet_sd(arch: str, units: int, seed: int, bn: bool = False, dims: int = 2) -> 'OrderedDict[str, np.ndarray]':
    from tools import synth_weights as sw
    sd = sw.resnet_sd_uncalibrated(arch, units, seed, bn, dims)
    return sw.calibrate_head(sd, score(arch, sd, sw.head_probe(seed, dims)))


def synthetic_basic_sd(sizes, units: int, seed: int, bn: bool = True) -> 'OrderedDict[str, np.ndarray]':
    from tools import synth_weights as sw
    return sw.basic_sd(sizes, units, seed, bn)
