"""Thin Python objects over the C-ABI: one Context per GPU, DeviceModel = a loaded layer program.

PyTorch is used only as plumbing: device memory (tensors), the current HIP stream and
torch.distributed.  All arithmetic of the hot path happens inside libtopaz_hip.so.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import TpzLayer, check, load_library

_contexts = {}


class Context:
    """tpz_ctx for one device.  Use get_context(device) rather than constructing directly."""

    def __init__(self, device: int = 0):
        if not torch.cuda.is_available():
            raise _lib.TopazHipError('no HIP device visible: topaz_amd needs an MI355X (gfx950); there is no CPU path')
        self.lib = load_library()
        self.device = int(device)
        h = C.c_void_p()
        check(self.lib.tpz_ctx_create(self.device, C.byref(h)))
        self.handle = h
        self._bound_stream = None

    def bind_current_stream(self) -> None:
        """run subsequent calls on torch's current stream for this device"""
        s = torch.cuda.current_stream(self.device).cuda_stream
        if s != self._bound_stream:
            check(self.lib.tpz_ctx_set_stream(self.handle, C.c_void_p(s)), self.handle)
            self._bound_stream = s

    def sync(self) -> None:
        check(self.lib.tpz_ctx_sync(self.handle), self.handle)

    def torch_device(self) -> torch.device:
        return torch.device('cuda', self.device)

    # ---- profiling counters (HIP events around every launch of a kernel class)
    def prof_enable(self, on=True) -> None:
        """True / 1: time every launch; 2: only convolution launches of >= 20 GFLOP; False / 0: off"""
        check(self.lib.tpz_prof_enable(self.handle, int(on)), self.handle)

    def prof_reset(self) -> None:
        check(self.lib.tpz_prof_reset(self.handle), self.handle)

    def set_exact(self, on: bool = True) -> None:
        """pin the fp32 MFMA kernels (no 2xf16 path) for models run on this context"""
        check(self.lib.tpz_ctx_set_exact(self.handle, 1 if on else 0), self.handle)

    def set_lanes(self, on=True) -> None:
        """patch lanes of tpz_denoise_2d / _3d (two auxiliary streams; an int 2 .. 4: that many); off: every launch on the ctx stream"""
        check(self.lib.tpz_ctx_set_lanes(self.handle, int(on)), self.handle)

    def set_tiling(self, limit_px: int = 40 << 20, tile: int = 4096) -> None:
        """internal tiling of the scoring pass: images above `limit_px` pixels are scored in tile^2 tiles with a receptive-field halo"""
        check(self.lib.tpz_ctx_set_tiling(self.handle, int(limit_px), int(tile)), self.handle)

    def set_raster(self, on: bool = True) -> None:
        """patch raster of the large 8-wave conv launches: each XCD walks 8 x 4 blocks of neighbouring tiles (default on)"""
        check(self.lib.tpz_ctx_set_raster(self.handle, 1 if on else 0), self.handle)

    def set_range(self, on: bool = True) -> None:
        """range scaling of the scoring pass: run on x * 2^-s with the biases scaled alike when max|x| > 32 (default on)"""
        check(self.lib.tpz_ctx_set_range(self.handle, 1 if on else 0), self.handle)

    def set_batch(self, n: int = 8) -> None:
        """patches / tiles per batched launch of tpz_denoise_2d / _3d on the 2xf16 path (0: off, the patch lanes instead)"""
        check(self.lib.tpz_ctx_set_batch(self.handle, int(n)), self.handle)

    def set_batch_memory(self, nbytes: int = 0) -> None:
        """device bytes a batched denoise pass may take for its per-image workspaces (0: 90 % of the free memory)"""
        check(self.lib.tpz_ctx_set_batch_memory(self.handle, int(nbytes)), self.handle)

    def launches(self) -> int:
        """kernel launches issued on this context so far (convolutions and elementwise)"""
        return int(self.lib.tpz_prof_launches(self.handle))

    def set_roi(self, on: bool = True) -> None:
        """patch windows of tpz_denoise_2d: each layer of a patch computes only what the kept centre depends on (default on)"""
        check(self.lib.tpz_ctx_set_roi(self.handle, 1 if on else 0), self.handle)

    def set_rw(self, on: bool = True) -> None:
        """the weights-resident kernel for the 3x3 32 -> 32 layers of the 32-unit detectors (csrc/conv_rw.h; default on)"""
        check(self.lib.tpz_ctx_set_rw(self.handle, 1 if on else 0), self.handle)

    def set_persist(self, mode: int = 1, workgroups: int = 0) -> None:
        """persistent workgroups of the 2xf16 convolutions: 0 never, 1 large launches (default), 2 every eligible launch with
        `workgroups` workgroups (0: one per grid slot)"""
        check(self.lib.tpz_ctx_set_persist(self.handle, int(mode), int(workgroups)), self.handle)

    def prof_get(self, cls: int) -> Tuple[float, int, float]:
        ms, n, fl = C.c_double(), C.c_longlong(), C.c_double()
        check(self.lib.tpz_prof_get(self.handle, cls, C.byref(ms), C.byref(n), C.byref(fl)), self.handle)
        return ms.value, n.value, fl.value


def _prof_get_dominant(self):
    """(name, total ms, launches, algorithmic FLOP) of the conv_mfma instantiation with the most time"""
    ms, n, fl = C.c_double(), C.c_longlong(), C.c_double()
    buf = C.create_string_buffer(256)
    check(self.lib.tpz_prof_get_dominant(self.handle, C.byref(ms), C.byref(n), C.byref(fl), buf, 256), self.handle)
    return buf.value.decode(), ms.value, n.value, fl.value


def _prof_kernels(self):
    """[(name, total ms, launches, FLOP)] of every conv_mfma instantiation launched since the last reset,
    by accumulated time"""
    out = []
    while True:
        ms, n, fl = C.c_double(), C.c_longlong(), C.c_double()
        buf = C.create_string_buffer(256)
        check(self.lib.tpz_prof_get_kernel(self.handle, len(out), C.byref(ms), C.byref(n), C.byref(fl), buf, 256),
              self.handle)
        if not buf.value:
            return out
        out.append((buf.value.decode(), ms.value, n.value, fl.value))


def _prof_kernels_bytes(self):
    """[(name, total ms, launches, FLOP, algorithmic HBM bytes)]: prof_kernels() plus the bytes each instantiation's launches
    read and wrote once (inputs with halo, weights, outputs) -- the bandwidth of the HBM-bound kernels"""
    out = []
    for rank, (name, ms, n, fl) in enumerate(self.prof_kernels()):
        b = C.c_double()
        check(self.lib.tpz_prof_get_kernel_bytes(self.handle, rank, C.byref(b)), self.handle)
        out.append((name, ms, n, fl, b.value))
    return out


def _mfma_sustained(self, ms: int = 300, zero_operands: bool = False):
    """(dense f16 TFLOP/s, s_memtime / s_memrealtime tick ratio) of a register-resident v_mfma_f32_16x16x32_f16 loop on every SIMD"""
    tf, ratio = C.c_double(), C.c_double()
    check(self.lib.tpz_prof_mfma_sustained(self.handle, int(ms), int(bool(zero_operands)), C.byref(tf), C.byref(ratio)), self.handle)
    return tf.value, ratio.value


Context.mfma_sustained = _mfma_sustained
Context.prof_get_dominant = _prof_get_dominant
Context.prof_kernels = _prof_kernels
Context.prof_kernels_bytes = _prof_kernels_bytes


def get_context(device: Optional[int] = None) -> Context:
    if device is None or device < 0:
        device = torch.cuda.current_device() if torch.cuda.is_available() else 0
    key = device
    ctx = _contexts.get(key)
    if ctx is None:
        ctx = Context(device)
        _contexts[key] = ctx
    ctx.bind_current_stream()
    return ctx


def _ptr(t: torch.Tensor) -> C.c_void_p:
    return C.c_void_p(t.data_ptr())


def as_device_f32(x, ctx: Context) -> torch.Tensor:
    """numpy / tensor -> contiguous fp32 tensor on the ctx device"""
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x))
    return x.to(device=ctx.torch_device(), dtype=torch.float32).contiguous()


class Stage:
    """tpz_stage: ring of (pinned host buffer, device buffer) slots with a copy stream of its own.  The H2D copy of the
    next image and the D2H copy of the previous result run under the current image's kernels (include/topaz_hip.h)."""

    def __init__(self, ctx: Context, slot_bytes: int, depth: int = 2):
        self.ctx, self.slot_bytes, self.depth = ctx, int(slot_bytes), int(depth)
        h = C.c_void_p()
        check(ctx.lib.tpz_stage_create(ctx.handle, self.slot_bytes, self.depth, C.byref(h)), ctx.handle)
        self.handle = h

    def close(self) -> None:
        if self.handle:
            self.ctx.lib.tpz_stage_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def host_array(self, slot: int, shape, dtype=np.float32) -> np.ndarray:
        """numpy view of slot's pinned buffer (fill it, then upload())"""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        assert n <= self.slot_bytes
        p = self.ctx.lib.tpz_stage_host_ptr(self.handle, slot)
        buf = (C.c_char * n).from_address(p)
        return np.frombuffer(buf, dtype=dtype).reshape(shape)

    def device_tensor(self, slot: int, shape) -> torch.Tensor:
        """fp32 tensor aliasing slot's device buffer (valid between acquire() and release())"""
        n = int(np.prod(shape))
        assert 4 * n <= self.slot_bytes
        p = self.ctx.lib.tpz_stage_device_ptr(self.handle, slot)
        holder = _DevPtr(p, 4 * n)
        return torch.as_tensor(holder, device=self.ctx.torch_device()).view(torch.float32).reshape(shape)

    def upload(self, slot: int, nbytes: int, src: Optional[np.ndarray] = None) -> None:
        ptr = None if src is None else np.ascontiguousarray(src).ctypes.data_as(C.c_void_p)
        check(self.ctx.lib.tpz_stage_h2d(self.handle, slot, ptr, int(nbytes)), self.ctx.handle)

    def acquire(self, slot: int) -> None:
        self.ctx.bind_current_stream()
        check(self.ctx.lib.tpz_stage_acquire(self.handle, slot), self.ctx.handle)

    def release(self, slot: int) -> None:
        check(self.ctx.lib.tpz_stage_release(self.handle, slot), self.ctx.handle)

    def download(self, slot: int, src: torch.Tensor) -> None:
        check(self.ctx.lib.tpz_stage_d2h(self.handle, slot, _ptr(src), src.numel() * src.element_size()), self.ctx.handle)

    def wait(self, slot: int) -> None:
        check(self.ctx.lib.tpz_stage_wait(self.handle, slot), self.ctx.handle)


class _DevPtr:
    """__cuda_array_interface__ holder for a raw device pointer owned by the library"""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {'shape': (nbytes,), 'typestr': '|u1', 'data': (int(ptr), False), 'version': 3}


def score_host(model: 'DeviceModel', x: np.ndarray) -> np.ndarray:
    """tpz_score_2d_host: numpy image in, numpy logits out (synchronous; staging inside the library)"""
    x = np.ascontiguousarray(x, dtype=np.float32)
    H, W = x.shape
    model.ctx.bind_current_stream()
    Do, Ho, Wo = model.out_shape(1, H, W)
    y = np.empty((Ho, Wo), dtype=np.float32)
    check(model.ctx.lib.tpz_score_2d_host(model.handle, x.ctypes.data_as(C.c_void_p), H, W, y.ctypes.data_as(C.c_void_p)),
          model.ctx.handle)
    return y


def denoise_host(model: 'DeviceModel', x: np.ndarray, patch: int, pad: int) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    H, W = x.shape
    model.ctx.bind_current_stream()
    y = np.empty_like(x)
    check(model.ctx.lib.tpz_denoise_2d_host(model.handle, x.ctypes.data_as(C.c_void_p), H, W, int(patch), int(pad),
                                            y.ctypes.data_as(C.c_void_p)), model.ctx.handle)
    return y


def nms_host(x: np.ndarray, r: int, threshold: float, ctx: Optional[Context] = None):
    """tpz_nms_2d_host: numpy score map in, (scores, coords) numpy arrays out"""
    ctx = ctx or get_context()
    ctx.bind_current_stream()
    x = np.ascontiguousarray(x, dtype=np.float32)
    H, W = x.shape
    cap = x.size if r < 1 else max(1024, min(x.size, (4 * x.size) // max(1, r * r) + 1024))
    thr = float(threshold)
    if thr == float('-inf'):
        thr = -3.4028234663852886e38 * 2
    while True:
        coords = np.empty((cap, 2), dtype=np.int32)
        scores = np.empty((cap,), dtype=np.float32)
        n = C.c_int(0)
        rc = ctx.lib.tpz_nms_2d_host(ctx.handle, x.ctypes.data_as(C.c_void_p), H, W, int(r), thr,
                                     coords.ctypes.data_as(C.c_void_p), scores.ctypes.data_as(C.c_void_p), cap, C.byref(n))
        if rc != 0 and n.value > cap:
            cap = n.value
            continue
        check(rc, ctx.handle)
        return scores[:n.value].copy(), coords[:n.value].copy()


class LayerProgram:
    """Host-side builder of the tpz_layer list + weight blob (the model manifest)."""

    def __init__(self, dims: int = 2):
        self.dims = dims
        self.layers: List[TpzLayer] = []
        self.blob: List[np.ndarray] = []
        self.n_floats = 0
        self.n_slots = 1          # slot 0 = input

    def new_slot(self) -> int:
        s = self.n_slots
        self.n_slots += 1
        return s

    def _push(self, a) -> int:
        a = np.ascontiguousarray(np.asarray(a, dtype=np.float32).ravel())
        off = self.n_floats
        self.blob.append(a)
        self.n_floats += a.size
        return off

    def conv(self, src: int, weight, bias=None, dil: int = 1, pad: int = 0, slope: float = 1.0, src2: int = -1,
             res: int = -1, res_crop: int = 0, post_scale=None, post_shift=None, head_w=None, head_b=None,
             dst: Optional[int] = None) -> int:
        w = np.asarray(weight, dtype=np.float32)
        cout, cin, k = w.shape[0], w.shape[1], w.shape[-1]
        L = TpzLayer()
        L.op = _lib.TPZ_OP_CONV
        L.dims = self.dims
        L.src, L.src2 = src, src2
        L.dst = self.new_slot() if dst is None else dst
        L.cin, L.cout, L.k, L.dil, L.pad = cin, cout, k, dil, pad
        L.slope = float(slope)
        L.w_off = self._push(w)
        L.b_off = self._push(bias) if bias is not None else -1
        L.res, L.res_crop = res, res_crop
        if post_scale is not None:
            L.post_scale_off = self._push(post_scale)
            L.post_shift_off = self._push(post_shift)
        else:
            L.post_scale_off = L.post_shift_off = -1
        if head_w is not None:
            L.head = 1
            L.head_w_off = self._push(head_w)
            L.head_b_off = self._push(np.asarray([head_b], dtype=np.float32))
        else:
            L.head = 0
            L.head_w_off = L.head_b_off = -1
        self.layers.append(L)
        return L.dst

    def maxpool2(self, src: int) -> int:
        L = TpzLayer()
        L.op = _lib.TPZ_OP_MAXPOOL2
        L.dims = self.dims
        L.src, L.src2, L.res = src, -1, -1
        L.dst = self.new_slot()
        L.w_off = L.b_off = L.post_scale_off = L.post_shift_off = L.head_w_off = L.head_b_off = -1
        L.slope = 1.0
        self.layers.append(L)
        return L.dst

    def maxpool(self, src: int, k: int = 3, dil: int = 1) -> int:
        """k^dims max over a window dilated by `dil`, stride 1, no padding (TPZ_OP_MAXPOOL: a filled MaxPool(k, stride))"""
        L = TpzLayer()
        L.op = _lib.TPZ_OP_MAXPOOL
        L.dims = self.dims
        L.src, L.src2, L.res = src, -1, -1
        L.dst = self.new_slot()
        L.k, L.dil, L.pad = k, dil, 0
        L.w_off = L.b_off = L.post_scale_off = L.post_shift_off = L.head_w_off = L.head_b_off = -1
        L.slope = 1.0
        self.layers.append(L)
        return L.dst

    def flat_blob(self) -> np.ndarray:
        if not self.blob:
            return np.zeros(1, dtype=np.float32)
        return np.concatenate(self.blob).astype(np.float32, copy=False)


class DeviceModel:
    """tpz_model: weights packed into MFMA fragment order and resident in HBM."""

    def __init__(self, program: LayerProgram, ctx: Optional[Context] = None):
        self.ctx = ctx or get_context()
        self.dims = program.dims
        lib = self.ctx.lib
        n = len(program.layers)
        arr = (TpzLayer * n)(*program.layers)
        blob = program.flat_blob()
        h = C.c_void_p()
        check(lib.tpz_model_load(self.ctx.handle, arr, n, blob.ctypes.data_as(C.c_void_p), blob.size, C.byref(h)),
              self.ctx.handle)
        self.handle = h
        n_conv, n_split, off = self.split_layers()
        if 0 < n_split < n_conv:
            import warnings
            warnings.warn(f'topaz_amd: {n_conv - n_split} of {n_conv} convolution layers of this model have no 2xf16 kernel and run '
                          f'on the fp32 matrix-core kernels (correct, several times slower): {off}', RuntimeWarning, stacklevel=3)

    def split_layers(self) -> Tuple[int, int, str]:
        """(convolution layers, those on the 2xf16 path, description of the others)"""
        a, b = C.c_int(), C.c_int()
        buf = C.create_string_buffer(512)
        check(self.ctx.lib.tpz_model_split_layers(self.handle, C.byref(a), C.byref(b), buf, 512), self.ctx.handle)
        return a.value, b.value, buf.value.decode()

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                self.ctx.lib.tpz_model_free(self.handle)
                self.handle = None
        except Exception:
            pass

    def out_shape(self, D: int, H: int, W: int) -> Tuple[int, int, int]:
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        check(self.ctx.lib.tpz_model_out_shape(self.handle, D, H, W, C.byref(a), C.byref(b), C.byref(c)),
              self.ctx.handle)
        return a.value, b.value, c.value

    def split_stats(self) -> Tuple[bool, int, int]:
        """(eligible for the 2xf16 path, images finished on it, images re-run on the fp32 kernels)"""
        e, a, b = C.c_int(), C.c_longlong(), C.c_longlong()
        check(self.ctx.lib.tpz_model_split_stats(self.handle, C.byref(e), C.byref(a), C.byref(b)), self.ctx.handle)
        return bool(e.value), a.value, b.value

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x: [N,1,(D,)H,W] (or without the channel axis) on the ctx device -> [N,1,(Do,)Ho,Wo]"""
        self.ctx.bind_current_stream()
        nd = self.dims
        if x.dim() == nd + 1:
            x = x.unsqueeze(1)
        if x.dim() != nd + 2 or x.shape[1] != 1:
            raise ValueError(f'expected [N,1,{"D,H,W" if nd == 3 else "H,W"}] input, got {tuple(x.shape)}')
        x = as_device_f32(x, self.ctx)
        N = x.shape[0]
        D = x.shape[2] if nd == 3 else 1
        H, W = x.shape[-2], x.shape[-1]
        Do, Ho, Wo = self.out_shape(D, H, W)
        if min(Do, Ho, Wo) < 1:
            raise ValueError(f'input {tuple(x.shape)} is too small for this model')
        co = C.c_int(1)
        check(self.ctx.lib.tpz_model_out_channels(self.handle, C.byref(co)), self.ctx.handle)
        shape = (N, co.value, Do, Ho, Wo) if nd == 3 else (N, co.value, Ho, Wo)
        y = torch.empty(shape, dtype=torch.float32, device=x.device)
        check(self.ctx.lib.tpz_model_forward(self.handle, _ptr(x), N, D, H, W, _ptr(y)), self.ctx.handle)
        return y

    def denoise_2d(self, x: torch.Tensor, patch: int, pad: int) -> torch.Tensor:
        self.ctx.bind_current_stream()
        x = as_device_f32(x, self.ctx)
        H, W = x.shape
        y = torch.empty_like(x)
        check(self.ctx.lib.tpz_denoise_2d(self.handle, _ptr(x), H, W, int(patch), int(pad), _ptr(y)), self.ctx.handle)
        return y

    def denoise_3d(self, x: torch.Tensor, patch: int, pad: int, shard: int = 0, n_shards: int = 1) -> torch.Tensor:
        """n_shards > 1: only this shard's tiles are denoised; the rest of the returned volume is zero (sum the shards)"""
        self.ctx.bind_current_stream()
        x = as_device_f32(x, self.ctx)
        D, H, W = x.shape
        y = torch.empty_like(x) if n_shards == 1 else torch.zeros_like(x)
        check(self.ctx.lib.tpz_denoise_3d_shard(self.handle, _ptr(x), D, H, W, int(patch), int(pad), int(shard), int(n_shards),
                                                _ptr(y)), self.ctx.handle)
        return y


# ---- single ops ------------------------------------------------------------------------------------
def conv(x: torch.Tensor, weight, bias=None, dil: int = 1, pad: int = 0, slope: float = 1.0,
         x2: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None, res_crop: int = 0,
         post_scale=None, post_shift=None, head_w=None, head_b: float = 0.0,
         ctx: Optional[Context] = None) -> torch.Tensor:
    """One fused convolution through tpz_conv.  x: [C1,(D1,)H1,W1]; x2 (optional, concatenated after the
    nearest-upsampled x): [C2,(D,)H,W].  Returns [Cout,(Do,)Ho,Wo] (or [1,...] with a fused head)."""
    ctx = ctx or get_context()
    ctx.bind_current_stream()
    w = np.ascontiguousarray(np.asarray(weight, dtype=np.float32))
    dims = w.ndim - 2
    x = as_device_f32(x, ctx)
    c1 = x.shape[0]
    D1 = x.shape[1] if dims == 3 else 1
    H1, W1 = x.shape[-2], x.shape[-1]
    if x2 is not None:
        x2 = as_device_f32(x2, ctx)
        D = x2.shape[1] if dims == 3 else 1
        H, W = x2.shape[-2], x2.shape[-1]
        cin = c1 + x2.shape[0]
    else:
        D, H, W, cin = D1, H1, W1, c1
    cout, k = w.shape[0], w.shape[-1]
    assert w.shape[1] == cin, (w.shape, cin)
    span = dil * (k - 1)
    Do = D + 2 * pad - span if dims == 3 else 1
    Ho, Wo = H + 2 * pad - span, W + 2 * pad - span
    co = 1 if head_w is not None else cout
    shape = (co, Do, Ho, Wo) if dims == 3 else (co, Ho, Wo)
    y = torch.empty(shape, dtype=torch.float32, device=x.device)

    def hp(a):
        if a is None:
            return None, C.c_void_p(None)
        a = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
        return a, a.ctypes.data_as(C.c_void_p)

    kb, pb = hp(bias)
    ks, ps_ = hp(post_scale)
    kt, pt = hp(post_shift)
    kh, ph = hp(head_w)
    if res is not None:
        res = as_device_f32(res, ctx)
    check(ctx.lib.tpz_conv(ctx.handle, dims, _ptr(x), c1, D1, H1, W1, _ptr(x2) if x2 is not None else C.c_void_p(None),
                           cin, D, H, W, w.ctypes.data_as(C.c_void_p), pb, cout, k, dil, pad, float(slope),
                           _ptr(res) if res is not None else C.c_void_p(None), res_crop, ps_, pt, ph, float(head_b),
                           _ptr(y)), ctx.handle)
    return y


def conv_split(x: torch.Tensor, weight, bias=None, dil: int = 1, pad: int = 0, slope: float = 1.0,
               res: Optional[torch.Tensor] = None, res_crop: int = 0, post_scale=None, post_shift=None,
               head_w=None, head_b: float = 0.0, ctx: Optional[Context] = None) -> Tuple[torch.Tensor, bool]:
    """The same 2-D convolution on the 2xf16 kernels (tpz_conv_split_2d): fp32 tensors in and out, converted
    to split f16 cells on the device.  Returns (y, overflow) -- overflow: a result left the f16 range."""
    ctx = ctx or get_context()
    ctx.bind_current_stream()
    w = np.ascontiguousarray(np.asarray(weight, dtype=np.float32))
    x = as_device_f32(x, ctx)
    cin, H, W = x.shape
    cout, k = w.shape[0], w.shape[-1]
    assert w.ndim == 4 and w.shape[1] == cin, (w.shape, cin)
    span = dil * (k - 1)
    Ho, Wo = H + 2 * pad - span, W + 2 * pad - span
    y = torch.empty((1 if head_w is not None else cout, Ho, Wo), dtype=torch.float32, device=x.device)

    def hp(a):
        if a is None:
            return None, C.c_void_p(None)
        a = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
        return a, a.ctypes.data_as(C.c_void_p)

    kb, pb = hp(bias)
    ks, ps_ = hp(post_scale)
    kt, pt = hp(post_shift)
    kh, ph = hp(head_w)
    if res is not None:
        res = as_device_f32(res, ctx)
    ovf = C.c_int(0)
    check(ctx.lib.tpz_conv_split_2d(ctx.handle, _ptr(x), cin, H, W, w.ctypes.data_as(C.c_void_p), pb, cout, k, dil,
                                    pad, float(slope), _ptr(res) if res is not None else C.c_void_p(None), res_crop,
                                    ps_, pt, ph, float(head_b), _ptr(y), C.byref(ovf)), ctx.handle)
    return y, bool(ovf.value)


def gmm_fit(x, pis, splits, alpha: float = 900.0, beta: float = 1.0, scale: float = 1.0, num_iters: int = 100,
            tol: float = 1e-3, ctx: Optional[Context] = None):
    """tpz_gmm_fit: EM fits of the 2-component pixel mixture for every initialisation (topaz/stats.py:87-203).
    x: pixel values (any array / device tensor); returns (mus, stds, pis, logps) as float64 arrays."""
    ctx = ctx or get_context()
    ctx.bind_current_stream()
    if not torch.is_tensor(x):
        x = torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=np.float32)))
    x = as_device_f32(x.reshape(-1), ctx)
    pis = np.ascontiguousarray(np.asarray(pis, dtype=np.float64))
    splits = np.ascontiguousarray(np.asarray(splits, dtype=np.float64))
    n = len(pis)
    outs = [np.zeros(n, dtype=np.float64) for _ in range(4)]
    check(ctx.lib.tpz_gmm_fit(ctx.handle, _ptr(x), x.numel(), pis.ctypes.data_as(C.c_void_p),
                              splits.ctypes.data_as(C.c_void_p), n, float(alpha), float(beta), float(scale),
                              int(num_iters), float(tol), *[o.ctypes.data_as(C.c_void_p) for o in outs]), ctx.handle)
    return tuple(outs)


def maxpool2(x: torch.Tensor, ctx: Optional[Context] = None) -> torch.Tensor:
    ctx = ctx or get_context()
    ctx.bind_current_stream()
    x = as_device_f32(x, ctx)
    dims = x.dim() - 1
    Cc = x.shape[0]
    D = x.shape[1] if dims == 3 else 1
    H, W = x.shape[-2], x.shape[-1]
    shape = (Cc, D // 2, H // 2, W // 2) if dims == 3 else (Cc, H // 2, W // 2)
    y = torch.empty(shape, dtype=torch.float32, device=x.device)
    check(ctx.lib.tpz_maxpool2(ctx.handle, dims, _ptr(x), Cc, D, H, W, _ptr(y)), ctx.handle)
    return y


def mean_std(x: torch.Tensor, unbiased: bool, ctx: Optional[Context] = None) -> Tuple[float, float]:
    ctx = ctx or get_context()
    ctx.bind_current_stream()
    x = as_device_f32(x, ctx)
    out = (C.c_float * 2)()
    check(ctx.lib.tpz_mean_std(ctx.handle, _ptr(x), x.numel(), 1 if unbiased else 0, out), ctx.handle)
    return float(out[0]), float(out[1])


def affine(x: torch.Tensor, scale: float, shift: float, ctx: Optional[Context] = None) -> torch.Tensor:
    ctx = ctx or get_context()
    ctx.bind_current_stream()
    x = as_device_f32(x, ctx)
    y = torch.empty_like(x)
    check(ctx.lib.tpz_affine(ctx.handle, _ptr(x), x.numel(), float(scale), float(shift), _ptr(y)), ctx.handle)
    return y


def normalize(x: torch.Tensor, mean: float, std: float, ctx: Optional[Context] = None) -> torch.Tensor:
    """(x - mean) / std as numpy evaluates it in fp32 (tpz_normalize): subtraction first, IEEE division; std == 0 -> inf / nan"""
    ctx = ctx or get_context()
    ctx.bind_current_stream()
    x = as_device_f32(x, ctx)
    y = torch.empty_like(x)
    check(ctx.lib.tpz_normalize(ctx.handle, _ptr(x), x.numel(), float(mean), float(std), _ptr(y)), ctx.handle)
    return y


def filter_2d(x: torch.Tensor, w, bias: float = 0.0, ctx: Optional[Context] = None) -> torch.Tensor:
    ctx = ctx or get_context()
    ctx.bind_current_stream()
    x = as_device_f32(x, ctx)
    w = np.ascontiguousarray(np.asarray(w, dtype=np.float32))
    H, W = x.shape
    y = torch.empty_like(x)
    check(ctx.lib.tpz_filter_2d(ctx.handle, _ptr(x), H, W, w.ctypes.data_as(C.c_void_p), w.shape[-1], float(bias),
                                _ptr(y)), ctx.handle)
    return y


def nms(score: torch.Tensor, r: int, threshold: float, scale: float = 1.0, ctx: Optional[Context] = None,
        cap: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """device NMS.  score [H,W] or [D,H,W]; returns (scores[n] fp32, coords[n,dims] int32 (x,y[,z])) on device"""
    ctx = ctx or get_context()
    ctx.bind_current_stream()
    score = as_device_f32(score, ctx)
    dims = score.dim()
    n_el = score.numel()
    if cap is None:
        # a pick suppresses at least itself; with r >= 1 picks are > r apart, so this bound is generous
        cap = n_el if r < 1 else max(1024, min(n_el, (4 * n_el) // max(1, r * r) + 1024))
    while True:
        coords = torch.empty((cap, dims), dtype=torch.int32, device=score.device)
        out = torch.empty((cap,), dtype=torch.float32, device=score.device)
        n = C.c_int(0)
        thr = float(threshold)
        if thr == float('-inf'):
            thr = -3.4028234663852886e38 * 2   # passes through c_float as -inf
        if dims == 2:
            H, W = score.shape
            rc = ctx.lib.tpz_nms_2d(ctx.handle, _ptr(score), H, W, int(r), thr, _ptr(coords), _ptr(out), cap,
                                    C.byref(n))
        else:
            D, H, W = score.shape
            rc = ctx.lib.tpz_nms_3d(ctx.handle, _ptr(score), D, H, W, int(r), float(scale), thr, _ptr(coords),
                                    _ptr(out), cap, C.byref(n))
        if rc != 0 and n.value > cap:
            cap = n.value          # capacity guess too small: rerun with the exact size
            continue
        check(rc, ctx.handle)
        return out[:n.value], coords[:n.value]


def transpose(x: torch.Tensor, ctx: Optional[Context] = None) -> torch.Tensor:
    ctx = ctx or get_context()
    ctx.bind_current_stream()
    x = as_device_f32(x, ctx)
    R, Cc = x.shape
    y = torch.empty((Cc, R), dtype=torch.float32, device=x.device)
    check(ctx.lib.tpz_transpose_2d(ctx.handle, _ptr(x), R, Cc, _ptr(y)), ctx.handle)
    return y
