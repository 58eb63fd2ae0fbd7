"""MRC reader / writer, restating the on-disk format handled by topaz/mrc.py (header struct
:11-107, parse :109-129, dtype modes :138-170, make_header :173-198, write :205-238).

Header: one native-endian C struct of exactly 1024 bytes; the image follows the `next` bytes of
extended header; modes 0/1/2/3/4/6/12/16; data is `nz*ny*nx` elements in C order, squeezed to 2-D
when nz == 1; `write` always stores float32 and forces mode 2 on a supplied header.
"""
from __future__ import annotations

import struct
from collections import namedtuple
from typing import Any, Tuple

import numpy as np

# (struct code, field names) in file order
_FIELDS = [
    ('3i', 'nx ny nz'), ('i', 'mode'), ('3i', 'nxstart nystart nzstart'), ('3i', 'mx my mz'),
    ('3f', 'xlen ylen zlen'), ('3f', 'alpha beta gamma'), ('3i', 'mapc mapr maps'),
    ('3f', 'amin amax amean'), ('2i', 'ispg next'), ('h', 'creatid'), ('30x', ''), ('2h', 'nint nreal'),
    ('20x', ''), ('2i', 'imodStamp imodFlags'), ('6h', 'idtype lens nd1 nd2 vd1 vd2'),
    ('6f', 'tilt_ox tilt_oy tilt_oz tilt_cx tilt_cy tilt_cz'), ('3f', 'xorg yorg zorg'),
    ('4s', 'cmap'), ('4s', 'stamp'), ('f', 'rms'), ('i', 'nlabl'), ('800s', 'labels'),
]
header_struct = struct.Struct(''.join(code for code, _ in _FIELDS))
assert header_struct.size == 1024
MRCHeader = namedtuple('MRCHeader', ' '.join(names for _, names in _FIELDS if names))

_MODE_TO_DTYPE = {0: np.int8, 1: np.int16, 2: np.float32, 3: '2h', 4: np.complex64, 6: np.uint16, 12: np.float16,
                  16: '3B'}


def parse_header(header_bytes: bytes) -> MRCHeader:
    return MRCHeader._make(header_struct.unpack(header_bytes))


def get_mode_from_header(header):
    try:
        return _MODE_TO_DTYPE[header.mode]
    except KeyError:
        raise Exception('Unknown dtype mode:' + str(header.mode))


def get_mode_for_header(dtype):
    for mode, dt in _MODE_TO_DTYPE.items():
        if mode != 12 and np.dtype(dtype) == np.dtype(dt):
            return mode
    raise ValueError('MRC incompatible dtype: ' + str(dtype))


def parse(content: bytes) -> Tuple[np.ndarray, Any, Any]:
    header = parse_header(content[:1024])
    start = 1024 + header.next
    extended_header = content[1024:start]
    array = np.frombuffer(content[start:], dtype=get_mode_from_header(header))
    array = array[:header.nz * header.ny * header.nx]
    array = np.reshape(array, (header.nz, header.ny, header.nx))
    if header.nz == 1:
        array = array[0]
    return array, header, extended_header


def read_into(path: str, alloc):
    """(image, header, extended header) of an MRC file whose pixels are real scalars (modes 0, 1, 2, 6, 12), decoded as float32
    straight into the array `alloc(shape)` returns -- a pinned staging buffer: a float32 file is read INTO that memory (one
    copy from the page cache instead of read + frombuffer + astype + copy), the integer / half modes are converted into it.
    Returns None for the vector modes (3, 4, 16): the caller takes the general `parse` route."""
    with open(path, 'rb') as f:
        header = parse_header(f.read(1024))
        mode = get_mode_from_header(header)
        if isinstance(mode, str) or np.dtype(mode).kind == 'c':
            return None
        extended_header = f.read(header.next)
        dt = np.dtype(mode)
        n = header.nz * header.ny * header.nx
        shape = (header.ny, header.nx) if header.nz == 1 else (header.nz, header.ny, header.nx)
        out = alloc(shape)
        if dt == np.float32:
            got = f.readinto(memoryview(out.reshape(-1)).cast('B'))
            if got != 4 * n:
                raise ValueError(f'{path}: {got} bytes of image data, header says {4 * n}')
        else:
            raw = np.frombuffer(f.read(n * dt.itemsize), dtype=dt)
            if raw.size != n:
                raise ValueError(f'{path}: {raw.size} pixels, header says {n}')
            np.copyto(out, raw.reshape(shape), casting='unsafe')
    return out, header, extended_header


def make_header(shape, cella, cellb, mz=1, dtype=np.float32, order=(1, 2, 3), dmin=0, dmax=-1, dmean=-2, rms=-1,
                exthd_size=0, ispg=0) -> MRCHeader:
    return MRCHeader(shape[2], shape[1], shape[0], get_mode_for_header(dtype), 0, 0, 0, 1, 1, mz,
                     cella[0], cella[1], cella[2], cellb[0], cellb[1], cellb[2], 1, 2, 3, dmin, dmax, dmean,
                     ispg, exthd_size, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                     b'\x00' * 4, b'\x00' * 4, rms, 0, b'\x00' * 800)


def write(f, array, header=None, extended_header=b'', ax=1, ay=1, az=1, alpha=0, beta=0, gamma=0):
    array = np.ascontiguousarray(array, dtype=np.float32)          # (no copy when it already is)
    if extended_header is None:
        extended_header = b''
    if header is None:
        header = make_header(array.shape, (ax, ay, az), (alpha, beta, gamma), mz=1, dmin=array.min(), dmax=array.max(),
                             dmean=array.mean(), rms=array.std(), exthd_size=len(extended_header))
    else:
        header = header._replace(mode=2)          # only the mode is refreshed (mrc.py:231-232)
    f.write(header_struct.pack(*list(header)))
    f.write(extended_header)
    f.write(memoryview(array.reshape(-1)).cast('B'))               # (the array's own memory: no tobytes() copy)
