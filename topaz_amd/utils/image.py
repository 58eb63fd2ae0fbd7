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
