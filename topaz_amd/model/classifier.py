"""Host-side mirror of topaz.model.classifier.LinearClassifier (classifier.py:14-66) over the
feature extractors of topaz/model/features/{resnet,basic}.py, for INFERENCE in filled mode.

The object keeps the reference's surface -- .width, .latent_dim, .eval(), .fill(), .unfill(),
.cuda(), .__call__(x) -- but owns no torch modules: the forward pass is one tpz_model_forward
call into libtopaz_hip.so.  Only the filled (dense, stride-1, dilated) network is implemented,
which is the only mode the extract / segment pipelines use (extract.py:229-231).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Optional

import numpy as np
import torch

from . import pack
from ..runtime import DeviceModel, get_context


class Features:
    """stand-in for model.features: exposes width / latent_dim like ResNet / BasicConv do"""

    def __init__(self, arch: str, units: int, bn: bool, width: int):
        self.arch, self.units, self.bn, self.width = arch, units, bn, width
        self.latent_dim = 4 * units if arch.startswith('resnet') else units
        self.dims = 2


class LinearClassifier:
    def __init__(self, arch: str, state_dict, dims: int = 2):
        if dims != 2:
            raise NotImplementedError('3-D scoring networks are not part of the MI355X hot path yet')
        self.arch = arch
        self.dims = dims
        self.state_dict_np = OrderedDict((k, (v.detach().cpu().numpy() if hasattr(v, 'detach') else np.asarray(v)))
                                         for k, v in state_dict.items())
        if arch in ('resnet8', 'resnet16'):
            self._program, width = pack.pack_resnet(arch, self.state_dict_np)
            units = self.state_dict_np['features.features.0.conv.weight'].shape[0]
            bn = any(k.endswith('running_mean') for k in self.state_dict_np)
        elif arch in pack.BASIC_SIZES:
            self._program, width = pack.pack_basicconv(pack.BASIC_SIZES[arch], self.state_dict_np)
            units = self.state_dict_np['features.features.0.weight'].shape[0]
            bn = any(k.endswith('running_mean') for k in self.state_dict_np)
        else:
            raise ValueError(f'unsupported feature extractor {arch!r}')
        self.features = Features(arch, units, bn, width)
        self.filled = False
        self._device_model: Optional[DeviceModel] = None
        self._device: Optional[int] = None

    # ---- reference surface
    @property
    def width(self) -> int:
        return self.features.width

    @property
    def latent_dim(self) -> int:
        return self.features.latent_dim

    def state_dict(self):
        return OrderedDict((k, torch.from_numpy(np.array(v))) for k, v in self.state_dict_np.items())

    def eval(self):
        return self

    def train(self, mode: bool = True):
        if mode:
            raise NotImplementedError('topaz_amd implements inference only')
        return self

    def fill(self, stride: int = 1) -> int:
        """LinearClassifier.fill -> ResNet.fill / BasicConv.fill: returns the total stride"""
        self.filled = True
        if self.arch.startswith('resnet'):
            return stride * pack.resnet_fill(pack.resnet_modules(self.arch))
        return stride * 2 ** (len(pack.BASIC_SIZES[self.arch]) - 1)

    def unfill(self):
        self.filled = False

    def cuda(self, device: Optional[int] = None):
        ctx = get_context(device)
        # keyed by the context (device AND lane): a tpz_ctx's workspace pool and flags are not thread-safe, so two lane
        # threads sharing one classifier must not end up on the same context
        if self._device_model is None or self._device_model.ctx is not ctx:
            self._device_model = DeviceModel(self._program, ctx)
            self._device = ctx.device
        return self

    @property
    def device_model(self) -> DeviceModel:
        if self._device_model is None:
            self.cuda()
        return self._device_model

    def to(self, device):
        d = torch.device(device)
        if d.type != 'cuda':
            raise RuntimeError('topaz_amd models run on the MI355X only (no CPU path)')
        return self.cuda(d.index)

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        return self.forward(x)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not self.filled:
            raise NotImplementedError('only the filled network (model.fill()) is implemented on the MI355X path; '
                                      'the strided training-time forward is out of scope')
        if self._device_model is None:
            self.cuda(x.device.index if x.is_cuda else None)
        return self._device_model.forward(x)
