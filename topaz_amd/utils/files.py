"""Pick-table writers, restating topaz/utils/files.py write_table (:242-268), write_via_csv (:109-146),
topaz/utils/conversions.py coordinates_to_boxes (:83-97), coordinates_to_eman2_json (:131-139),
coordinates_to_star (:173-192) and topaz/utils/star.py write (:91-98)."""
from __future__ import annotations

import json

import numpy as np
import pandas as pd

_STAR_NAMES = {'score': 'AutopickFigureOfMerit', 'image_name': 'MicrographName', 'x_coord': 'CoordinateX',
               'y_coord': 'CoordinateY', 'voltage': 'Voltage', 'detector_pixel_size': 'DetectorPixelSize',
               'magnification': 'Magnification', 'amplitude_contrast': 'AmplitudeContrast'}


def coordinates_to_boxes(coords, box_width, box_height):
    x, y = coords[:, 0], coords[:, 1]
    bw = np.array([box_width] * len(x), dtype=np.int32)
    bh = np.array([box_height] * len(x), dtype=np.int32)
    return np.stack([x - bw // 2, y - bh // 2, bw, bh], 1)


def coordinates_to_star(table, image_ext=''):
    table = table.copy()
    for k, v in _STAR_NAMES.items():
        if k in table.columns:
            table[v] = table[k]
            table = table.drop(k, axis=1)
    table['MicrographName'] = table['MicrographName'].apply(lambda x: x + image_ext)
    return table


def write_star(table, f):
    print('data_images', file=f)
    print('loop_', file=f)
    for i, name in enumerate(table.columns):
        print('_rln' + name + ' #' + str(i + 1), file=f)
    table.to_csv(f, sep='\t', index=False, header=False)


def write_via_csv(path, table):
    """VGG Image Annotator region CSV, one point region per pick (utils/files.py:109-146 upstream): per row the image as
    `<name>.png`, the number of picks of that image and the pick's running index within it, the point as a JSON shape and
    the score as a JSON attribute."""
    per_image = table.groupby('image_name', sort=False)['image_name']
    point = '{{"name":"point","cx":{},"cy":{}}}'.format
    rows = {
        'filename': table['image_name'].astype(str) + '.png',
        'file_size': -1,
        'file_attributes': '{}',
        'region_count': per_image.transform('size').to_numpy(),
        'region_id': per_image.cumcount().to_numpy(),
        'region_shape_attributes': [point(x, y) for x, y in zip(table['x_coord'], table['y_coord'])],
        'region_attributes': ['{{"score":"{}"}}'.format(v) for v in table['score']] if 'score' in table.columns else '{}',
    }
    pd.DataFrame(rows, index=table.index).to_csv(path, index=False)


def write_table(f, table, format='auto', boxsize=0, image_ext=''):
    if format == 'box':
        xy = table[['x_coord', 'y_coord']].values.astype(np.int32)
        pd.DataFrame(coordinates_to_boxes(xy, boxsize, boxsize)).to_csv(f, sep='\t', header=False, index=False)
    elif format == 'json':
        xy = table[['x_coord', 'y_coord']].values.astype(int)
        json.dump({'boxes': [[int(x), int(y), 'manual'] for x, y in xy]}, f, indent=0)
    elif format == 'star':
        write_star(coordinates_to_star(table, image_ext=image_ext), f)
    elif format == 'csv':
        write_via_csv(f, table)
    else:
        columns = ['image_name', 'x_coord', 'y_coord']
        if 'score' in table.columns:
            columns.append('score')
        table[columns].to_csv(f, sep='\t', index=False)


def format_pick_rows(name: str, coords, scores, dims: int = 2) -> bytes:
    """the rows `name\tx\ty[\tz]\tscore\n` of a pick table exactly as the reference's f-string prints them
    (topaz/extract.py:341-354: a float32 score prints with the digits of its float64 value), formatted by the library's
    host-side writer (tpz_format_picks) -- ~60 ns per row where a Python loop needs ~2 us: at 24 k picks per 4096^2
    micrograph that loop cost more than the GPU work of the micrograph."""
    import ctypes as C
    from .._lib import load_library
    scores = np.ascontiguousarray(scores, dtype=np.float32).reshape(-1)
    n = int(scores.shape[0])
    if n == 0:
        return b''
    coords = np.ascontiguousarray(np.asarray(coords).reshape(n, -1), dtype=np.int32)
    nm = name.encode()
    cap = n * (len(nm) + 80)
    buf = C.create_string_buffer(cap)
    got = load_library().tpz_format_picks(nm, coords.ctypes.data_as(C.c_void_p), int(coords.shape[1]), int(dims),
                                          scores.ctypes.data_as(C.c_void_p), n, C.cast(buf, C.c_void_p), cap)
    if got < 0:
        raise RuntimeError(f'tpz_format_picks failed ({got})')
    return buf.raw[:got]
