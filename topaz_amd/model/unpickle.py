"""Read the weights and hyper-parameters out of a full-module pickle written by the reference
(`torch.save(model, path)`, topaz/training.py:302,601; loaded by factory.py:55 with
torch.load(weights_only=False)) WITHOUT importing the reference package.

The pickle references topaz.model.classifier.LinearClassifier, topaz.model.features.resnet.{ResNet8,
ResNet16,BasicConv,ResidA,MaxPool}, topaz.model.features.basic.{BasicConv,Conv127,...} and
torch.nn.modules.* (SURVEY.md P10).  The Unpickler maps every `topaz.*` global onto an inert stand-in
class that just records its attribute dict, lets through an explicit allowlist of torch / numpy /
collections globals (tensor reconstruction, storages, torch.nn layer classes) and refuses everything
else with UnpicklingError -- `torch.load(weights_only=False)` alone would execute arbitrary globals.
The architecture is then recognised from the recorded class names and the state_dict is rebuilt
by walking the `_modules` / `_parameters` / `_buffers` dictionaries.
"""
from __future__ import annotations

import pickle
from collections import OrderedDict

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


# Globals a module pickle written by torch.save(model) legitimately needs (pickletools dump of the
# reference's own files): tensor / parameter reconstruction, storages, dtypes, torch.nn layer classes, containers.
# Everything else -- os.system, builtins.eval, subprocess ... -- is refused: a model file is data, not a program.
_ALLOWED_EXACT = {
    ('collections', 'OrderedDict'), ('builtins', 'set'), ('__builtin__', 'set'), ('builtins', 'frozenset'),
    ('builtins', 'slice'), ('builtins', 'complex'), ('builtins', 'bytearray'),
    ('torch._utils', '_rebuild_tensor'), ('torch._utils', '_rebuild_tensor_v2'), ('torch._utils', '_rebuild_parameter'),
    ('torch._utils', '_rebuild_parameter_with_state'), ('torch._utils', '_rebuild_qtensor'),
    ('torch', 'Size'), ('torch', 'device'), ('torch', 'dtype'),
    ('torch.serialization', '_get_layout'), ('torch.nn.parameter', 'Parameter'), ('torch.nn.parameter', 'Buffer'),
    ('numpy.core.multiarray', '_reconstruct'), ('numpy._core.multiarray', '_reconstruct'),
    ('numpy.core.multiarray', 'scalar'), ('numpy._core.multiarray', 'scalar'), ('numpy', 'ndarray'), ('numpy', 'dtype'),
}
_ALLOWED_TORCH_SUFFIX = ('Storage', 'Tensor')        # torch.FloatStorage, torch.LongStorage, torch.FloatTensor, ...
_TORCH_DTYPES = {'float16', 'float32', 'float64', 'bfloat16', 'int8', 'int16', 'int32', 'int64', 'uint8', 'bool'}


def _allowed(module: str, name: str) -> bool:
    if (module, name) in _ALLOWED_EXACT:
        return True
    if module == 'torch' and (name.endswith(_ALLOWED_TORCH_SUFFIX) or name in _TORCH_DTYPES):
        return True
    if module == 'torch.storage' and name in ('UntypedStorage', 'TypedStorage'):
        return True
    # layer classes: plain attribute containers, instantiated via __reduce__ / __setstate__ only
    if module.startswith('torch.nn.modules.') and name[:1].isupper() and name.isidentifier():
        return True
    return False


def _load_from_bytes(b):
    """torch.storage._load_from_bytes is torch.load(BytesIO(b), weights_only=False) with the STANDARD unpickler: a nested
    payload would escape the allowlist.  The same bytes through this module's own unpickler instead."""
    import io
    return torch.load(io.BytesIO(b), map_location='cpu', weights_only=False, pickle_module=_PickleModule)


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module == 'topaz' or module.startswith('topaz.'):
            return _stub_for(module, name)
        if (module, name) == ('torch.storage', '_load_from_bytes'):
            return _load_from_bytes
        if not _allowed(module, name):
            raise pickle.UnpicklingError(f'model file refers to {module}.{name}, which a topaz model pickle has no '
                                         f'business loading (allowlist: topaz_amd/model/unpickle.py)')
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


def _without_dropout(sd, kinds):
    """state_dict renumbered as if the nn.Dropout entries of features.features were not there (identity at inference;
    they only shift the index of every later module: resnet.py:296-303, basic.py:57-70)"""
    new_index, n = {}, 0
    for i, kind in enumerate(kinds):
        if kind != 'Dropout':
            new_index[str(i)] = str(n)
            n += 1
    out = OrderedDict()
    pre = 'features.features.'
    for k, v in sd.items():
        if k.startswith(pre):
            idx, _, rest = k[len(pre):].partition('.')
            k = f'{pre}{new_index[idx]}.{rest}'
        out[k] = v
    return out


def load_module_pickle(path, with_traits: bool = False):
    """(arch, state_dict) of a pickled LinearClassifier; with_traits=True adds {'pooling': the feature stack holds MaxPool
    layers (`topaz train --pooling max`, ResNet6), 'dropout': it was trained with --dropout > 0}.  The state_dict is
    numbered without the Dropout modules."""
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
    mods = feats.__dict__['_modules']['features'].__dict__['_modules']
    kinds = [type(m).__name__ for m in mods.values()]
    pooling = 'MaxPool' in kinds
    dropout = 'Dropout' in kinds
    if fname in ('ResNet6', 'ResNet8', 'ResNet16'):
        arch = fname.lower()
        if any(k not in ('BasicConv', 'ResidA', 'MaxPool', 'Dropout') for k in kinds):
            raise NotImplementedError(f'{path}: {arch} with layers {sorted(set(kinds))} is not supported on the MI355X path')
    elif fname in ('BasicConv', 'Conv127', 'Conv63', 'Conv31'):
        if any(not (k.startswith(('Conv', 'BatchNorm')) or k in ('PReLU', 'Dropout')) for k in kinds):
            raise NotImplementedError(f'{path}: BasicConv with layers {sorted(set(kinds))} is not supported on the MI355X '
                                      f'path (strided PReLU stacks only: no pooling, no other activation)')
        n_convs = sum(1 for k in kinds if k.startswith('Conv'))
        arch = {5: 'conv127', 4: 'conv63', 3: 'conv31'}.get(n_convs)
        if arch is None:
            raise ValueError(f'{path}: BasicConv with {n_convs} convolutions is not a conv127/63/31 stack')
    else:
        raise NotImplementedError(f'{path}: feature extractor {fname} is not supported on the MI355X path')
    if dropout:
        sd = _without_dropout(sd, kinds)
    return (arch, sd, {'pooling': pooling, 'dropout': dropout}) if with_traits else (arch, sd)
