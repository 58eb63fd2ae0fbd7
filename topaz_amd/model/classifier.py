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

    def __init__(self, arch: str, units: int, bn: bool, width: int, dims: int = 2):
        self.arch, self.units, self.bn, self.width = arch, units, bn, width
        self.latent_dim = 4 * units if arch.startswith('resnet') else units
        self.dims = dims


class LinearClassifier:
    def __init__(self, arch: str, state_dict, dims: Optional[int] = None, pooling: bool = False, dropout: bool = False):
        self.arch = arch
        self.dropout = bool(dropout)                 # trained with --dropout > 0 (matters to BasicConv.fill only, see pack.py)
        self.pooling = bool(pooling) or arch == 'resnet6'       # MaxPool(3, stride 2) layers (resnet.py:10-47)
        self.state_dict_np = OrderedDict((k, (v.detach().cpu().numpy() if hasattr(v, 'detach') else np.asarray(v)))
                                         for k, v in state_dict.items())
        # 2-D or 3-D (classifier.py:17-29 `dims`): read off the 1x1(x1) head unless given
        wdims = self.state_dict_np['classifier.weight'].ndim - 2
        if dims is None:
            dims = wdims
        if dims not in (2, 3) or dims != wdims:
            raise ValueError(f'LinearClassifier: dims = {dims} with {wdims}-D weights')
        self.dims = dims
        if arch in ('resnet6', 'resnet8', 'resnet16'):
            self._program, width = pack.pack_resnet(arch, self.state_dict_np, dims, self.pooling)
            units = self.state_dict_np['features.features.0.conv.weight'].shape[0]
            bn = any(k.endswith('running_mean') for k in self.state_dict_np)
        elif arch in pack.BASIC_SIZES:
            self._program, width = pack.pack_basicconv(pack.BASIC_SIZES[arch], self.state_dict_np, self.dropout, dims)
            units = self.state_dict_np['features.features.0.weight'].shape[0]
            bn = any(k.endswith('running_mean') for k in self.state_dict_np)
        else:
            raise ValueError(f'unsupported feature extractor {arch!r}')
        self.features = Features(arch, units, bn, width, dims)
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
            return stride * pack.resnet_fill(pack.resnet_modules(self.arch, self.pooling))
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


def classify_patches(classifier: LinearClassifier, tomo_stack, patch_size: int = 48, padding: int = 36, batch_size: int = 1,
                     volume_num: int = 1, total_volumes: int = 1, verbose: bool = True) -> torch.Tensor:
    """classify_patches (classifier.py:69-102): score a batch of tomograms tile by tile.  Tiles follow PatchDataset
    (denoising/datasets.py:412-468): a ceil(n / patch_size)^3 grid, each tile the (patch_size + 2 * padding)^3 crop
    around its cell with everything outside the volume left at ZERO, scored by the filled network (which pads again by
    width // 2), its centre pasted back.  Returns a tensor like `tomo_stack` ([N, D, H, W]) on the host.
    The tile assembly and the scoring run on the device; `batch_size` is accepted for compatibility (tiles are
    independent, one forward each)."""
    import sys
    stack = torch.as_tensor(tomo_stack)
    if stack.dim() != 4:
        raise ValueError(f'classify_patches expects a stack [N, D, H, W], got {tuple(stack.shape)}')
    if not classifier.filled:
        raise NotImplementedError('classify_patches needs the filled network (classifier.fill())')
    dm = classifier.device_model
    dev = dm.ctx.torch_device()
    out = torch.zeros(tuple(stack.shape), dtype=torch.float32)
    d = patch_size + 2 * padding
    for n in range(stack.shape[0]):
        tomo = stack[n].to(device=dev, dtype=torch.float32)
        Z, Y, X = tomo.shape
        scored = torch.zeros((Z, Y, X), dtype=torch.float32, device=dev)
        cells = [(i, j, k) for i in range(0, Z, patch_size) for j in range(0, Y, patch_size) for k in range(0, X, patch_size)]
        tile = torch.empty((1, 1, d, d, d), dtype=torch.float32, device=dev)
        for count, (i, j, k) in enumerate(cells, 1):
            tile.zero_()
            lo = [max(0, v - padding) for v in (i, j, k)]
            hi = [min(n_, v + patch_size + padding) for n_, v in zip((Z, Y, X), (i, j, k))]
            off = [padding - v + l for v, l in zip((i, j, k), lo)]
            tile[0, 0, off[0]:off[0] + hi[0] - lo[0], off[1]:off[1] + hi[1] - lo[1], off[2]:off[2] + hi[2] - lo[2]] = \
                tomo[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]]
            y = classifier(tile)[0, 0]
            pz, py, px = min(patch_size, Z - i), min(patch_size, Y - j), min(patch_size, X - k)
            scored[i:i + pz, j:j + py, k:k + px] = y[padding:padding + pz, padding:padding + py, padding:padding + px]
            if verbose:
                print(f'# [{volume_num}/{total_volumes}] {round(count * 100 / len(cells))}%', file=sys.stderr, end='\r')
        out[n] = scored.cpu()
    if verbose:
        print(' ' * 100, file=sys.stderr, end='\r')
    return out
