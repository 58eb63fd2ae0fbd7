"""Image loading / saving around the hot path, restating topaz/utils/data/loader.py:51-120
(load_mrc/load_tiff/load_png/load_jpeg/load_pil/load_image) and topaz/utils/image.py:88-147
(quantize/unquantize/save_image and the per-format writers)."""
from __future__ import annotations

import os

import numpy as np

from .. import mrc


def quantize(x, mi=-3, ma=3, dtype=np.uint8):
    if mi is None:
        mi = x.min()
    if ma is None:
        ma = x.max()
    x = 255 * (x - mi) / (ma - mi)
    return np.round(np.clip(x, 0, 255)).astype(dtype)


def unquantize(x, mi=-3, ma=3, dtype=np.float32):
    return x.astype(dtype) * (ma - mi) / 255 + mi


def load_mrc(path, standardize=False):
    with open(path, 'rb') as f:
        content = f.read()
    image, header, extended_header = mrc.parse(content)
    if image.dtype == np.float16:
        image = image.astype(np.float32)
    if standardize:
        image = image - header.amean
        image /= header.rms
    return image, header, extended_header


def load_pil(path, standardize=False):
    from PIL import Image
    im = Image.open(path)
    fp = im.fp
    im.load()
    if fp is not None:
        fp.close()
    x = np.array(im)
    if standardize and not (path.endswith('.png') or path.endswith('.jpg') or path.endswith('.jpeg')):
        x = (x - x.mean()) / x.std()
    # note: the reference's load_png / load_jpeg unquantize into a temporary and then return the PIL
    # image itself (loader.py:75-98), so 8-bit images reach the caller as their raw uint8 values.
    return x


def load_image(path, standardize=False, make_image=True, return_header=True):
    ext = os.path.splitext(path)[1]
    data = load_mrc(path, standardize) if ext == '.mrc' else load_pil(path, standardize)
    image, header, extended_header = data if type(data) == tuple else (data, None, None)
    if make_image:
        from PIL import Image
        image = Image.fromarray(image)
    return (image, header, extended_header) if (header and return_header) else image


def save_mrc(x, path, header=None, extended_header=None):
    with open(path, 'wb') as f:
        mrc.write(f, np.asarray(x)[np.newaxis], header=header, extended_header=extended_header or b'')


def save_image(x, path, mi=-3, ma=3, f=None, verbose=False, header=None, extended_header=None):
    if f is None:
        f = os.path.splitext(path)[1][1:]
    else:
        path = path + '.' + f
    if verbose:
        print('# saving:', path)
    if f == 'mrc':
        save_mrc(x, path, header=header, extended_header=extended_header)
    elif f in ('tiff', 'tif'):
        from PIL import Image
        Image.fromarray(x).save(path, 'tiff')
    elif f == 'png':
        from PIL import Image
        Image.fromarray(quantize(x, mi=mi, ma=ma)).save(path, 'png')
    elif f in ('jpg', 'jpeg'):
        from PIL import Image
        Image.fromarray(quantize(x, mi=mi, ma=ma)).save(path, 'jpeg')


# ---- truncated-DFT downsample (topaz/utils/image.py:38-61, `topaz downsample`) --------------------------------
# The reference computes rfft2 -> keep the m x (n//2+1) low-frequency block -> scale by (m*n)/(M*N) -> irfft2.
# That is a real-linear map of the image, separable into a complex row operator Lc (m x M) and, for the
# complex intermediate U = Lc x, two real column operators:  y = Re(U) R1 + Im(U) R2.  The operators are built
# once per shape in float64 with numpy's own FFT applied to identity matrices (so every convention -- which
# rows `F[-m//2:]` keeps for odd m, how irfft treats the Nyquist bin -- is inherited, not re-derived), and the
# image goes through two GEMMs on the fp32 MFMA kernel (1x1 convolutions) and two transposes on the device.
_DS_CACHE = {}


def _downsample_operators(M, N, m, n):
    key = (M, N, m, n)
    if key not in _DS_CACHE:
        F = np.fft.fft(np.eye(M), axis=0)
        Lc = np.fft.ifft(np.concatenate([F[0:m // 2], F[-m // 2:]], axis=0), axis=0)        # m x M complex
        Lc = Lc * ((n * m) / (M * N))
        Fc = np.fft.fft(np.eye(N), axis=1)[:, 0:n // 2 + 1]                                  # N x K complex
        R1 = np.fft.irfft(Fc, n=n, axis=1)                                                   # N x n
        R2 = np.fft.irfft(1j * Fc, n=n, axis=1)
        L = np.concatenate([Lc.real, Lc.imag], axis=0).astype(np.float32)                    # 2m x M
        R = np.concatenate([R1, R2], axis=0).T.astype(np.float32)                            # n x 2N
        if len(_DS_CACHE) > 4:
            _DS_CACHE.clear()
        _DS_CACHE[key] = (np.ascontiguousarray(L), np.ascontiguousarray(R))
    return _DS_CACHE[key]


def _as_planes(t, channels, pixels):
    """[channels][pixels] matrix as a [C][H][W] tensor for the 1x1-conv GEMM (any H*W = pixels works)"""
    for w in (64, 32, 16):
        if pixels % w == 0:
            return t.reshape(channels, pixels // w, w)
    return t.reshape(channels, 1, pixels)


def downsample(x, factor=1, shape=None):
    """Downsample a 2-D array with a truncated DFT on the MI355X; numpy in, numpy out (dtype of x)."""
    import torch
    from .. import runtime as rt
    x = np.asarray(x)
    if x.ndim != 2:
        raise NotImplementedError('downsample: 2-D arrays only')
    M, N = x.shape
    if shape is None:
        shape = (int(M / factor), int(N / factor))
    m, n = shape
    L, R = _downsample_operators(M, N, m, n)
    xd = rt.as_device_f32(x, rt.get_context())
    U = rt.conv(_as_planes(xd, M, N), L.reshape(2 * m, M, 1, 1)).reshape(2 * m, N)           # [Re(U); Im(U)]
    Ut = rt.transpose(U)                                                                      # N x 2m
    # channels 0..N-1 read Re(U)^T (row stride 2m), channels N..2N-1 read Im(U)^T: feed both halves as one tensor
    Ucat = torch.cat([Ut[:, :m], Ut[:, m:]], 0).contiguous()                                  # 2N x m
    yt = rt.conv(_as_planes(Ucat, 2 * N, m), R.reshape(n, 2 * N, 1, 1)).reshape(n, m)
    return rt.transpose(yt).cpu().numpy().astype(x.dtype)


def downsample_file(path, scale, output, verbose=False):
    """topaz/utils/image.py:64-85"""
    import sys
    image = load_image(path, make_image=False)
    image, header, extended_header = image if type(image) is tuple else (image, None, None)
    image = image.astype(np.float32)
    small = downsample(image, scale)
    if header:
        header = header._replace(ny=small.shape[0], nx=small.shape[1])
    if verbose:
        print('Downsample image:', path, file=sys.stderr)
        print('From', image.shape, 'to', small.shape, file=sys.stderr)
    save_image(small, output, header=header, extended_header=extended_header)
    return small
