"""Mirror of topaz/model/factory.py:33-64 load_model: pretrained aliases -> architecture +
packaged state_dict, anything else -> a user-trained full-module pickle (topaz train output).

Pretrained state_dicts ship as package data under topaz_amd/pretrained/ (the same .sav files the
reference packages).  resnet8(_u64) / resnet16(_u64) are large blobs absent from the reference
checkout this build was made from; asking for them raises with that explanation.
"""
from __future__ import annotations

import os

import torch

from .classifier import LinearClassifier
from .unpickle import load_module_pickle

_PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'pretrained', 'detector')

_PRETRAINED = {
    'resnet16': ('resnet16', 'resnet16_u64.sav'),
    'resnet16_u64': ('resnet16', 'resnet16_u64.sav'),
    'resnet16_u32': ('resnet16', 'resnet16_u32.sav'),
    'resnet8': ('resnet8', 'resnet8_u64.sav'),
    'resnet8_u64': ('resnet8', 'resnet8_u64.sav'),
    'resnet8_u32': ('resnet8', 'resnet8_u32.sav'),
}


def load_state_dict_from_pkg(name: str):
    path = os.path.join(_PKG, name)
    if not os.path.exists(path):
        raise RuntimeError(f'Could not load resource topaz_amd/pretrained/detector/{name}: the blob is not packaged '
                           f'(it is missing from the reference checkout as well); train or supply the model file')
    return torch.load(path, map_location='cpu', weights_only=True)


def load_model(path):
    if isinstance(path, LinearClassifier):
        return path
    if path in _PRETRAINED:
        arch, name = _PRETRAINED[path]
        return LinearClassifier(arch, load_state_dict_from_pkg(name))
    # user model: torch.save(model) of LinearClassifier(ResNet*/BasicConv) (factory.py:54-56, training.py:601)
    arch, state_dict, traits = load_module_pickle(path, with_traits=True)
    return LinearClassifier(arch, state_dict, pooling=traits['pooling'], dropout=traits['dropout'])
