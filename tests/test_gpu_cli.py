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
    # the whole map against the TIFF the reference's segment_images wrote for this micrograph
    ref = np.array(Image.open(os.path.join(CLI, 'segment_mic_a.tiff')))
    assert ref.shape == y.shape and np.abs(y - ref).max() <= 1e-4
    # patched mode (`-p`): upstream crashes on an unsupported keyword; the evident intent is tiles of
    # 2 * patch_size with width // 2 halo, whose stitched map equals the whole-image one
    _run(['segment', '-m', 'resnet8_u32', '-p', '48', '-o', str(tmp_path / 'segp'), str(tmp_path / 'mic_a.mrc')])
    yp = np.array(Image.open(tmp_path / 'segp' / 'mic_a.tiff'))
    assert np.abs(yp - ref).max() <= 1e-4


def _capture(argv):
    import contextlib
    import io
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        _run(argv)
    return buf.getvalue()


def _radius_lines(text):
    rows = []
    for line in text.strip().split('\n'):
        if line.startswith('# radius='):
            rows.append({k: float(v) for k, v in (kv.split('=') for kv in line[2:].split(', '))})
    return rows


def test_extract_targets_radius_search_and_validation(gpu_ctx, tmp_path, monkeypatch):
    """`topaz extract --targets`: the radius search (no -r) prints one line per radius and extracts at the optimum; with -r it
    prints the validation line.  Compared with the stdout and the pick file of the reference's own CLI on the same
    micrographs and targets table (oracle/make_golden.py extras): same radii, same recall / target counts, AUPRC and RMSE
    to 1e-6, the same pick table."""
    for n in ('mic_a.mrc', 'mic_b.mrc'):
        shutil.copy(os.path.join(CLI, n), tmp_path / n)
    monkeypatch.chdir(tmp_path)                      # the targets table names the micrographs as the command line does
    out = tmp_path / 'opt.txt'
    text = _capture(['extract', '-m', 'resnet8_u32', '-d', '0', '--targets', os.path.join(CLI, 'targets.txt'), '--min-radius', '4',
                     '--max-radius', '16', '--step-radius', '4', '-o', str(out), 'mic_a.mrc', 'mic_b.mrc'])
    got, ref = _radius_lines(text), _radius_lines(open(os.path.join(CLI, 'targets_search_stdout.txt')).read())
    assert [g['radius'] for g in got] == [4, 8, 12, 16] and len(got) == len(ref)
    for g, r in zip(got, ref):
        assert (g['radius'], g['recall'], g['targets']) == (r['radius'], r['recall'], r['targets'])
        assert abs(g['auprc'] - r['auprc']) <= 1e-6 and abs(g['rmse'] - r['rmse']) <= 1e-6
    a, b = pd.read_csv(out, sep='\t'), pd.read_csv(os.path.join(CLI, 'targets_search_picks.txt'), sep='\t')
    assert a.image_name.tolist() == b.image_name.tolist()
    assert np.array_equal(a[['x_coord', 'y_coord']].values, b[['x_coord', 'y_coord']].values)
    assert np.abs(a.score.values - b.score.values).max() <= 1e-4
    text = _capture(['extract', '-m', 'resnet8_u32', '-d', '0', '-r', '8', '--assignment-radius', '5', '--targets',
                     os.path.join(CLI, 'targets.txt'), '--only-validate', 'mic_a.mrc', 'mic_b.mrc'])
    (g,), (r,) = _radius_lines(text), _radius_lines(open(os.path.join(CLI, 'targets_validate_stdout.txt')).read())
    assert (g['radius'], g['recall'], g['targets']) == (r['radius'], r['recall'], r['targets'])
    assert abs(g['auprc'] - r['auprc']) <= 1e-6 and abs(g['rmse'] - r['rmse']) <= 1e-6


def test_extract_tomogram_dims3(gpu_ctx, tmp_path):
    """`topaz extract --dims 3`: a 3-D classifier pickle scores a tomogram read from an MRC file and the 3-D suppression writes
    x, y, z columns; the table equals the reference's non_maximum_suppression_3d of the reference's own score map wherever
    the device map agrees with it (the golden's threshold sits at the median of a dense map: ties within 1e-4 may flip)"""
    from conftest import load_golden
    from topaz_amd import mrc
    z = load_golden('score_resnet8_3d_u8')
    with open(tmp_path / 'tomo.mrc', 'wb') as f:
        mrc.write(f, z['x0'])
    out = tmp_path / 'picks3d.txt'
    _run(['extract', '-m', os.path.join(GOLDEN, 'user_model_resnet8_3d_u8.sav'), '--dims', '3', '-r', str(int(z['nms_r'])), '-t',
          repr(float(z['nms_thr'])), '-d', '0', '-o', str(out), str(tmp_path / 'tomo.mrc')])
    t = pd.read_csv(out, sep='\t')
    assert list(t.columns) == ['image_name', 'x_coord', 'y_coord', 'z_coord', 'score']
    ref_c, ref_s = z['nms_coords'], z['nms_scores']
    got = {tuple(r) for r in t[['x_coord', 'y_coord', 'z_coord']].values.tolist()}
    want = {tuple(r) for r in ref_c.tolist()}
    assert len(got ^ want) <= max(2, len(want) // 50), (len(got), len(want), len(got ^ want))
    common = sorted(got & want)
    lookup = {tuple(c): s for c, s in zip(ref_c.tolist(), ref_s.tolist())}
    mine = {tuple(r[:3]): r[3] for r in t[['x_coord', 'y_coord', 'z_coord', 'score']].values.tolist()}
    assert max(abs(mine[tuple(map(float, c))] - lookup[c]) for c in common) <= 1e-4


def test_single_rank_rccl_first_contact(gpu_ctx):
    """Everything of the multi-GPU path that one GPU can exercise over the real backend: a 1-rank 'nccl' (= RCCL) process group
    with a bound device, the size all_gather + the ONE packed gather of pick tables on device tensors, the tomogram reduce, the
    barrier with device ids, the pre-flight report -- the calls that would otherwise meet RCCL for the first time on the 8-GPU
    node (run in a child process: a process group cannot be re-created in the test process)."""
    import subprocess
    import sys
    code = '''
import os, sys, io
sys.path.insert(0, %r)
import torch
from topaz_amd import parallel
os.environ.update(WORLD_SIZE='1', RANK='0', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(parallel.free_port()),
                  TOPAZ_AMD_FORCE_DIST='1')
rank, local_rank, world = parallel.init_from_env()
assert (rank, local_rank, world) == (0, 0, 1) and torch.distributed.get_backend() == 'nccl'
dev = torch.device('cuda', 0)
buf = io.StringIO()
info = parallel.preflight(0, 0, 1, stream=buf)
assert 'device' in info and 'rank=0' in buf.getvalue()
parallel.barrier(dev)
assert parallel.sum_over_ranks(1.0, dev) == 1.0 and parallel.max_over_ranks(2.5, dev) == 2.5
assert parallel.gather_scalars(3.0, dev) == [3.0]
g = torch.Generator().manual_seed(1)
scores = [torch.randn(n, generator=g).cuda() for n in (5, 0, 17)]
coords = [torch.randint(0, 4096, (n, 2), generator=g, dtype=torch.int32).cuda() for n in (5, 0, 17)]
got = parallel.gather_pick_tables([10, 11, 12], scores, coords, dev)
assert sorted(got) == [10, 11, 12]
for i, s, c in zip((10, 11, 12), scores, coords):
    assert torch.equal(got[i][0], s.cpu()) and torch.equal(got[i][1], c.cpu())
vol = torch.randn(8, 16, 16, device=dev)
out = parallel.sum_to_root(vol.clone())
assert torch.equal(out, vol)
torch.distributed.destroy_process_group()
print('rccl-1-ok')
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and 'rccl-1-ok' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
