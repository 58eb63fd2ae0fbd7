"""ctypes binding of libtopaz_hip.so (include/topaz_hip.h).

There is deliberately no CPU fallback here: if the shared library is missing or no gfx950 device
is visible, the hot-path entry points raise.  (The CPU restatement of the reference lives under
oracle/ and is test infrastructure only.)
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libtopaz_hip.so')

TPZ_OP_CONV = 1
TPZ_OP_MAXPOOL2 = 2
TPZ_OP_MAXPOOL = 3


class TpzLayer(C.Structure):
    """struct tpz_layer of include/topaz_hip.h (field order and types must match)."""
    _fields_ = [
        ('op', C.c_int32), ('dims', C.c_int32), ('src', C.c_int32), ('src2', C.c_int32), ('dst', C.c_int32),
        ('cin', C.c_int32), ('cout', C.c_int32), ('k', C.c_int32), ('dil', C.c_int32), ('pad', C.c_int32),
        ('slope', C.c_float),
        ('w_off', C.c_int64), ('b_off', C.c_int64),
        ('res', C.c_int32), ('res_crop', C.c_int32),
        ('post_scale_off', C.c_int64), ('post_shift_off', C.c_int64),
        ('head', C.c_int32), ('reserved', C.c_int32),
        ('head_w_off', C.c_int64), ('head_b_off', C.c_int64),
    ]


_P = C.c_void_p
_SIGNATURES = {
    'tpz_version': (C.c_char_p, []),
    'tpz_last_error': (C.c_char_p, [_P]),
    'tpz_ctx_create': (C.c_int, [C.c_int, C.POINTER(_P)]),
    'tpz_ctx_destroy': (None, [_P]),
    'tpz_ctx_set_stream': (C.c_int, [_P, _P]),
    'tpz_ctx_sync': (C.c_int, [_P]),
    'tpz_model_load': (C.c_int, [_P, C.POINTER(TpzLayer), C.c_int, _P, C.c_size_t, C.POINTER(_P)]),
    'tpz_model_free': (None, [_P]),
    'tpz_model_forward': (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    'tpz_model_out_shape': (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                      C.POINTER(C.c_int)]),
    'tpz_model_out_channels': (C.c_int, [_P, C.POINTER(C.c_int)]),
    'tpz_denoise_2d': (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    'tpz_denoise_3d': (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    'tpz_denoise_3d_shard': (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    'tpz_mean_std': (C.c_int, [_P, _P, C.c_size_t, C.c_int, C.POINTER(C.c_float)]),
    'tpz_gmm_fit': (C.c_int, [_P, _P, C.c_size_t, _P, _P, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_double,
                              _P, _P, _P, _P]),
    'tpz_affine': (C.c_int, [_P, _P, C.c_size_t, C.c_float, C.c_float, _P]),
    'tpz_normalize': (C.c_int, [_P, _P, C.c_size_t, C.c_float, C.c_float, _P]),
    'tpz_format_picks': (C.c_longlong, [C.c_char_p, _P, C.c_int, C.c_int, _P, C.c_longlong, _P, C.c_longlong]),
    'tpz_filter_2d': (C.c_int, [_P, _P, C.c_int, C.c_int, _P, C.c_int, C.c_float, _P]),
    'tpz_nms_2d': (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_float, _P, _P, C.c_int, C.POINTER(C.c_int)]),
    'tpz_nms_3d': (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_float, _P, _P, C.c_int,
                             C.POINTER(C.c_int)]),
    'tpz_stage_create': (C.c_int, [_P, C.c_size_t, C.c_int, C.POINTER(_P)]),
    'tpz_stage_free': (None, [_P]),
    'tpz_stage_host_ptr': (_P, [_P, C.c_int]),
    'tpz_stage_device_ptr': (_P, [_P, C.c_int]),
    'tpz_stage_h2d': (C.c_int, [_P, C.c_int, _P, C.c_size_t]),
    'tpz_stage_acquire': (C.c_int, [_P, C.c_int]),
    'tpz_stage_release': (C.c_int, [_P, C.c_int]),
    'tpz_stage_d2h': (C.c_int, [_P, C.c_int, _P, C.c_size_t]),
    'tpz_stage_wait': (C.c_int, [_P, C.c_int]),
    'tpz_score_2d_host': (C.c_int, [_P, _P, C.c_int, C.c_int, _P]),
    'tpz_denoise_2d_host': (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    'tpz_nms_2d_host': (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_float, _P, _P, C.c_int, C.POINTER(C.c_int)]),
    'tpz_conv': (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int,
                           C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _P, C.c_int, _P, _P, _P,
                           C.c_float, _P]),
    'tpz_maxpool2': (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    'tpz_transpose_2d': (C.c_int, [_P, _P, C.c_int, C.c_int, _P]),
    'tpz_ctx_set_exact': (C.c_int, [_P, C.c_int]),
    'tpz_ctx_set_lanes': (C.c_int, [_P, C.c_int]),
    'tpz_ctx_set_batch': (C.c_int, [_P, C.c_int]),
    'tpz_ctx_set_batch_memory': (C.c_int, [_P, C.c_longlong]),
    'tpz_ctx_set_range': (C.c_int, [_P, C.c_int]),
    'tpz_ctx_set_raster': (C.c_int, [_P, C.c_int]),
    'tpz_ctx_set_tiling': (C.c_int, [_P, C.c_longlong, C.c_int]),
    'tpz_prof_launches': (C.c_longlong, [_P]),
    'tpz_ctx_set_roi': (C.c_int, [_P, C.c_int]),
    'tpz_ctx_set_rw': (C.c_int, [_P, C.c_int]),
    'tpz_ctx_set_persist': (C.c_int, [_P, C.c_int, C.c_int]),
    'tpz_model_split_stats': (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    'tpz_model_split_layers': (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_int]),
    'tpz_conv_split_2d': (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_float, _P, C.c_int, _P, _P, _P, C.c_float, _P, C.POINTER(C.c_int)]),
    'tpz_prof_enable': (C.c_int, [_P, C.c_int]),
    'tpz_prof_reset': (C.c_int, [_P]),
    'tpz_prof_get_dominant': (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.POINTER(C.c_double), C.c_char_p, C.c_int]),
    'tpz_prof_get_kernel': (C.c_int, [_P, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.POINTER(C.c_double), C.c_char_p, C.c_int]),
    'tpz_prof_get': (C.c_int, [_P, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.POINTER(C.c_double)]),
    'tpz_prof_get_kernel_bytes': (C.c_int, [_P, C.c_int, C.POINTER(C.c_double)]),
    'tpz_prof_mfma_sustained': (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    'tpz_debug_switches': (C.c_int, [C.c_char_p, C.c_int]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None
_lock = threading.Lock()


class TopazHipError(RuntimeError):
    pass


def load_library(path: str | None = None) -> C.CDLL:
    """dlopen libtopaz_hip.so and declare every prototype.  Raises if the library is missing."""
    global _lib
    # torch first: its wheel bundles its own HIP runtime (libamdhip64), and the two copies in one process do not share
    # devices -- loaded the other way round (library, then torch) tpz_ctx_create finds no device.  With torch's runtime
    # already mapped, the library's libamdhip64 dependency resolves to it.
    import torch  # noqa: F401
    with _lock:
        if _lib is not None and path is None:
            return _lib
        p = path or os.environ.get('TOPAZ_HIP_LIB') or LIB_PATH
        if not os.path.exists(p):
            raise TopazHipError(
                f'{p} not found: build it with `python -m topaz_amd.build` (hipcc, gfx950). '
                'topaz_amd has no CPU fallback for the hot path.')
        lib = C.CDLL(p)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if path is None:
            _lib = lib
        return lib


def check(rc: int, ctx_handle=None) -> None:
    if rc != 0:
        lib = load_library()
        msg = lib.tpz_last_error(ctx_handle)
        raise TopazHipError(msg.decode() if msg else f'libtopaz_hip error {rc}')
