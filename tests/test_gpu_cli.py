"""End to end through the CLI on the MI355X against files produced by the reference's own CLI / file
drivers (oracle/make_golden.py cli): MRC in -> pick tables / denoised MRC out."""
import os
import shutil

import numpy as np
import pandas as pd
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
CLI = os.path.join(GOLDEN, 'cli')


def _run(argv):
    from topaz_amd.main import main
    main(argv)


def test_extract_single_tsv(gpu_ctx, tmp_path):
    mics = []
    for n in ('mic_a.mrc', 'mic_b.mrc'):
        shutil.copy(os.path.join(CLI, n), tmp_path / n)
        mics.append(str(tmp_path / n))
    out = tmp_path / 'picks.txt'
    _run(['extract', '-m', 'resnet8_u32', '-r', '8', '-d', '0', '-o', str(out)] + mics)
    got = pd.read_csv(out, sep='\t')
    ref = pd.read_csv(os.path.join(CLI, 'extract_picks.txt'), sep='\t')
    assert list(got.columns) == list(ref.columns) and len(got) == len(ref)
    assert got.image_name.tolist() == ref.image_name.tolist()
    assert np.array_equal(got[['x_coord', 'y_coord']].values, ref[['x_coord', 'y_coord']].values)
    assert np.abs(got.score.values - ref.score.values).max() <= 1e-4


def test_extract_per_micrograph_star_with_scaling(gpu_ctx, tmp_path):
    shutil.copy(os.path.join(CLI, 'mic_a.mrc'), tmp_path / 'mic_a.mrc')
    _run(['extract', '-m', 'resnet8_u32', '-r', '8', '-t', '-3', '-x', '2', '-d', '0', '--per-micrograph', '--format',
          'star', '-o', str(tmp_path / 'out' / 'x'), str(tmp_path / 'mic_a.mrc')])
    got = open(tmp_path / 'out' / 'COORDS' / 'mic_a.star').read().strip().split('\n')
    ref = open(os.path.join(CLI, 'extract_mic_a.star')).read().strip().split('\n')
    assert got[:6] == ref[:6] and len(got) == len(ref)
    for g, r in zip(got[6:], ref[6:]):
        g, r = g.split('\t'), r.split('\t')
        assert g[1:] == r[1:] and abs(float(g[0]) - float(r[0])) <= 1e-4


def test_denoise_stream_mrc(gpu_ctx, tmp_path):
    from topaz_amd import mrc
    shutil.copy(os.path.join(CLI, 'mic_a.mrc'), tmp_path / 'mic_a.mrc')
    _run(['denoise', '-m', 'unet-v0.2.1', '-s', '96', '-p', '24', '-o', str(tmp_path / 'den'), str(tmp_path / 'mic_a.mrc')])
    got = open(tmp_path / 'den' / 'mic_a.mrc', 'rb').read()
    ref = open(os.path.join(CLI, 'denoise_mic_a.mrc'), 'rb').read()
    assert got[:1024] == ref[:1024]                       # header passes through, mode forced to 2
    a, _, _ = mrc.parse(got)
    b, _, _ = mrc.parse(ref)
    assert np.abs(a - b).max() <= 1e-4


def test_denoise3d_stream_mrc(gpu_ctx, tmp_path):
    from topaz_amd import mrc
    shutil.copy(os.path.join(CLI, 'tomo.mrc'), tmp_path / 'tomo.mrc')
    _run(['denoise3d', '-m', os.path.join(CLI, 'unet3d_nf8_state.sav'), '--base-kernel-width', '7', '-s', '32', '-p', '16',
          '-d', '0', '-o', str(tmp_path / 'den3'), str(tmp_path / 'tomo.mrc')])
    a, ha, _ = mrc.parse(open(tmp_path / 'den3' / 'tomo.mrc', 'rb').read())
    b, hb, _ = mrc.parse(open(os.path.join(CLI, 'denoise3d_tomo.mrc'), 'rb').read())
    assert np.abs(a - b).max() <= 1e-4
    assert (ha.nx, ha.ny, ha.nz, ha.mode) == (hb.nx, hb.ny, hb.nz, 2)
    assert abs(ha.amin - hb.amin) <= 1e-4 and abs(ha.amax - hb.amax) <= 1e-4 and abs(ha.amean - hb.amean) <= 1e-4


def test_segment_tiff(gpu_ctx, tmp_path):
    from PIL import Image
    from conftest import load_golden
    shutil.copy(os.path.join(CLI, 'mic_a.mrc'), tmp_path / 'mic_a.mrc')
    _run(['segment', '-m', 'resnet8_u32', '-o', str(tmp_path / 'seg'), str(tmp_path / 'mic_a.mrc')])
    y = np.array(Image.open(tmp_path / 'seg' / 'mic_a.tiff'))
    assert y.shape == (160, 200) and y.dtype == np.float32
    # consistency with the pick file of the reference: the top pick of mic_a is the arg-max of the map
    ref = pd.read_csv(os.path.join(CLI, 'extract_picks.txt'), sep='\t')
    top = ref[ref.image_name == 'mic_a'].iloc[0]
    assert abs(y[top.y_coord, top.x_coord] - top.score) <= 1e-4
