"""Read the weights and hyper-parameters out of a full-module pickle written by the reference
(`torch.save(model, path)`, topaz/training.py:302,601; loaded by factory.py:55 with
torch.load(weights_only=False)) WITHOUT importing the reference package.

The pickle references topaz.model.classifier.LinearClassifier, topaz.model.features.resnet.{ResNet8,
ResNet16,BasicConv,ResidA,MaxPool}, topaz.model.features.basic.{BasicConv,Conv127,...} and
torch.nn.modules.* (SURVEY.md P10).  A restricted Unpickler maps every `topaz.*` global onto an
inert stand-in class that just records its attribute dict; torch's own classes load normally.
The architecture is then recognised from the recorded class names and the state_dict is rebuilt
by walking the `_modules` / `_parameters` / `_buffers` dictionaries.
"""
from __future__ import annotations

import pickle
from collections import OrderedDict
from typing import Tuple

import torch


class _Stub:
    """inert stand-in for any class of the reference package"""
    _tpz_qualname = ''

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)


_stub_cache = {}


def _stub_for(module: str, name: str):
    key = f'{module}.{name}'
    if key not in _stub_cache:
        _stub_cache[key] = type(name, (_Stub,), {'_tpz_qualname': key})
    return _stub_cache[key]


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module == 'topaz' or module.startswith('topaz.'):
            return _stub_for(module, name)
        return super().find_class(module, name)


class _PickleModule:
    """the `pickle_module` interface torch.load expects"""
    __name__ = 'topaz_amd_unpickle'
    Unpickler = _Unpickler

    @staticmethod
    def load(f, **kw):
        return _Unpickler(f, **kw).load()


def _walk(mod, prefix, out):
    d = mod.__dict__
    for k, p in (d.get('_parameters') or {}).items():
        if p is not None:
            out[prefix + k] = p.detach().cpu()
    for k, b in (d.get('_buffers') or {}).items():
        if b is not None:
            out[prefix + k] = b.detach().cpu()
    for k, m in (d.get('_modules') or {}).items():
        if m is not None:
            _walk(m, prefix + k + '.', out)


def load_module_pickle(path) -> Tuple[str, 'OrderedDict[str, torch.Tensor]']:
    obj = torch.load(path, map_location='cpu', weights_only=False, pickle_module=_PickleModule)
    if isinstance(obj, (dict, OrderedDict)) and all(torch.is_tensor(v) for v in obj.values()):
        raise ValueError(f'{path} holds a bare state_dict; the architecture cannot be inferred. '
                         'Use a pretrained alias or a full-module file written by `topaz train`.')
    qn = getattr(type(obj), '_tpz_qualname', '')
    if not qn.endswith('LinearClassifier'):
        raise ValueError(f'{path}: expected a pickled topaz LinearClassifier, found {type(obj).__name__}')
    feats = obj.__dict__['_modules']['features']
    fq = type(feats)._tpz_qualname
    fname = fq.rsplit('.', 1)[-1]
    sd = OrderedDict()
    _walk(obj, '', sd)
    if fname in ('ResNet8', 'ResNet16'):
        arch = fname.lower()
        # pooling / ResNet6 variants change the graph; they are not on the hot path
        mods = feats.__dict__['_modules']['features'].__dict__['_modules']
        if any(type(m).__name__ == 'MaxPool' for m in mods.values()):
            raise NotImplementedError(f'{path}: ResNet with pooling layers is not supported on the MI355X path')
    elif fname in ('BasicConv', 'Conv127', 'Conv63', 'Conv31'):
        n_convs = sum(1 for k, v in sd.items() if k.startswith('features.features.') and v.dim() == 4)
        arch = {5: 'conv127', 4: 'conv63', 3: 'conv31'}.get(n_convs)
        if arch is None:
            raise ValueError(f'{path}: BasicConv with {n_convs} convolutions is not a conv127/63/31 stack')
    else:
        raise NotImplementedError(f'{path}: feature extractor {fname} is not supported on the MI355X path')
    return arch, sd
