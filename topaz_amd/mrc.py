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


def make_header(shape, cella, cellb, mz=1, dtype=np.float32, order=(1, 2, 3), dmin=0, dmax=-1, dmean=-2, rms=-1,
                exthd_size=0, ispg=0) -> MRCHeader:
    return MRCHeader(shape[2], shape[1], shape[0], get_mode_for_header(dtype), 0, 0, 0, 1, 1, mz,
                     cella[0], cella[1], cella[2], cellb[0], cellb[1], cellb[2], 1, 2, 3, dmin, dmax, dmean,
                     ispg, exthd_size, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                     b'\x00' * 4, b'\x00' * 4, rms, 0, b'\x00' * 800)


def write(f, array, header=None, extended_header=b'', ax=1, ay=1, az=1, alpha=0, beta=0, gamma=0):
    array = np.asarray(array).astype(np.float32)
    if extended_header is None:
        extended_header = b''
    if header is None:
        header = make_header(array.shape, (ax, ay, az), (alpha, beta, gamma), mz=1, dmin=array.min(), dmax=array.max(),
                             dmean=array.mean(), rms=array.std(), exthd_size=len(extended_header))
    else:
        header = header._replace(mode=2)          # only the mode is refreshed (mrc.py:231-232)
    f.write(header_struct.pack(*list(header)))
    f.write(extended_header)
    f.write(array.tobytes())
