"""Mirror of topaz/denoising/models.py load_model (:570-625) and the inference side of the
denoiser classes (UDenoiseNet :74-175, UDenoiseNetSmall :178-244, DenoiseNet2 :52-66,
UDenoiseNet3D :452-564) and topaz/filters.py AffineDenoise (:40-48).

A DenoiseNet owns the layer program + HBM-resident packed weights; `net(x)` is one
tpz_model_forward call.  Training (train_model etc.) is out of scope.
"""
from __future__ import annotations

import os
import sys
from collections import OrderedDict
from typing import Optional

import numpy as np
import torch

from ..model import pack
from ..runtime import DeviceModel, LayerProgram, get_context

_PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'pretrained', 'denoise')

# denoising/models.py:568-579
model_name_dict = {
    'unet': 'unet_L2_v0.2.2.sav',
    'unet-small': 'unet_small_L1_v0.2.2.sav',
    'fcnn': 'fcnn_L1_v0.2.2.sav',
    'affine': 'affine_L1_v0.2.2.sav',
    'unet-v0.2.1': 'unet_L2_v0.2.1.sav',
    'unet-3d': 'unet-3d-10a-v0.2.4.sav',
    'unet-3d-10a': 'unet-3d-10a-v0.2.4.sav',
    'unet-3d-20a': 'unet-3d-20a-v0.2.4.sav',
}

# file -> (kind, dims)   (architectures of denoising/models.py:593-609)
_ARCH = {
    'unet_L2_v0.2.1.sav': ('unet', 2), 'unet_L2_v0.2.2.sav': ('unet', 2), 'unet_small_L1_v0.2.2.sav': ('unet-small', 2),
    'fcnn_L1_v0.2.2.sav': ('fcnn', 2), 'affine_L1_v0.2.2.sav': ('affine', 2),
    'unet-3d-10a-v0.2.4.sav': ('unet-3d', 3), 'unet-3d-20a-v0.2.4.sav': ('unet-3d', 3),
}


def _program(kind: str, sd) -> LayerProgram:
    if kind == 'unet':
        return pack.pack_unet(sd, 5, 2)
    if kind == 'unet3':
        return pack.pack_unet(sd, 5, 2, noise_only=True)
    if kind == 'unet2':
        return pack.pack_unet(sd, 5, 2, no_skip=(2, 1))
    if kind == 'unet-small':
        return pack.pack_unet(sd, 3, 2)
    if kind == 'unet-3d':
        return pack.pack_unet(sd, 5, 3)
    if kind == 'fcnn':
        return pack.pack_fcnn(sd)
    if kind == 'affine':
        return pack.pack_filter(sd['filter.weight'], sd.get('filter.bias'))
    raise ValueError(kind)


def _kind_from_state_dict(sd) -> str:
    keys = set(sd)
    if 'filter.weight' in keys:
        return 'affine'
    if 'net.0.weight' in keys:
        return 'fcnn'
    if 'enc1.0.weight' in keys:
        dims = np.asarray(sd['enc1.0.weight']).ndim - 2
        if dims == 3:
            return 'unet-3d'
        return 'unet' if 'enc6.0.weight' in keys else 'unet-small'
    raise ValueError('unrecognised denoising state_dict')


class DenoiseNet:
    """inference-only stand-in for the reference's denoising nn.Modules"""

    def __init__(self, kind: str, state_dict):
        self.kind = kind
        self.state_dict_np = OrderedDict((k, (v.detach().cpu().numpy() if hasattr(v, 'detach') else np.asarray(v)))
                                         for k, v in state_dict.items())
        self.dims = 3 if kind == 'unet-3d' else 2
        self._program = _program(kind, self.state_dict_np)
        self._device_model: Optional[DeviceModel] = None

    def state_dict(self):
        return OrderedDict((k, torch.from_numpy(np.array(v))) for k, v in self.state_dict_np.items())

    def eval(self):
        return self

    def cuda(self, device: Optional[int] = None):
        ctx = get_context(device)
        if self._device_model is None or self._device_model.ctx is not ctx:
            self._device_model = DeviceModel(self._program, ctx)
        return self

    @property
    def device_model(self) -> DeviceModel:
        if self._device_model is None:
            self.cuda()
        return self._device_model

    def parameters(self):
        dev = self.device_model.ctx.torch_device()
        return iter([torch.empty(0, device=dev)])

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        return self.device_model.forward(x)


def load_model(name, base_kernel_width: int = 11) -> DenoiseNet:
    """aliases -> packaged state_dicts; otherwise a path: a bare state_dict (3-D models,
    models.py:619-622) or a full-module pickle (torch.save(model), models.py:628-633)."""
    if isinstance(name, DenoiseNet):
        return name
    pretrained = name in model_name_dict
    fname = model_name_dict.get(name, name)
    if fname in _ARCH:
        kind, _ = _ARCH[fname]
        path = os.path.join(_PKG, fname)
        if not os.path.exists(path):
            raise RuntimeError(f'Could not load resource topaz_amd/pretrained/denoise/{fname}: the blob is not '
                               f'packaged (missing from the reference checkout as well)')
        print('# loading pretrained model:', fname, file=sys.stderr)
        return DenoiseNet(kind, torch.load(path, map_location='cpu', weights_only=True))
    # user file
    from ..model.unpickle import _PickleModule, _walk
    obj = torch.load(fname, map_location='cpu', weights_only=False, pickle_module=_PickleModule)
    if isinstance(obj, (dict, OrderedDict)):
        sd = obj                      # bare state_dict (how the reference saves 3-D models): the keys decide
        return DenoiseNet(_kind_from_state_dict(sd), sd)
    sd = OrderedDict()
    _walk(obj, '', sd)
    return DenoiseNet(_kind_from_class(obj, sd, fname), sd)


# classes of topaz/denoising/models.py (and filters.AffineDenoise) this path evaluates -> kind.  The state_dict keys
# alone do not identify the forward pass: UDenoiseNet3 (--arch unet3, models.py:339-449) has UDenoiseNet's keys but
# returns x - dec1(h), UDenoiseNet2 / DenoiseNet differ in their layer lists.
_CLASS_KIND = {
    'UDenoiseNet': 'unet',
    'UDenoiseNetSmall': 'unet-small',
    'UDenoiseNet2': 'unet2',          # --arch unet2: no skip connection into dec2 / dec1 (models.py:247-336)
    'UDenoiseNet3': 'unet3',          # --arch unet3: predicts the noise, returns x - dec1(h) (models.py:339-449)
    'DenoiseNet2': 'fcnn',
    'UDenoiseNet3D': 'unet-3d',
    'AffineDenoise': 'affine',
}
# what the parameter names alone look like for each class (several classes share a layout)
_KEY_LAYOUT = {'unet2': 'unet', 'unet3': 'unet'}


def _kind_from_class(obj, sd, path) -> str:
    qn = getattr(type(obj), '_tpz_qualname', '') or type(obj).__name__
    cls = qn.rsplit('.', 1)[-1]
    if cls == 'DenoiseNet':
        # --arch fcnet: its Sequential feeds a 3*nf-channel tensor into Conv2d(nf, 2*nf) (models.py:38-39) -- the forward pass
        # raises RuntimeError upstream for every width, so no such model can have been trained
        raise NotImplementedError(f'{path}: topaz.denoising.models.DenoiseNet (--arch fcnet) cannot be evaluated: its layer '
                                  f'list is inconsistent upstream (denoising/models.py:25-49, channel mismatch at net.10)')
    if cls not in _CLASS_KIND:
        raise NotImplementedError(f'{path}: denoising model class {cls} is not supported on the MI355X path '
                                  f'(supported: {", ".join(sorted(_CLASS_KIND))})')
    kind = _CLASS_KIND[cls]
    layout = _kind_from_state_dict(sd)
    if layout != _KEY_LAYOUT.get(kind, kind):
        raise ValueError(f'{path}: a {cls} whose parameters look like a {layout!r} network')
    return kind
