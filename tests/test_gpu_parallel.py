"""The RCCL code path on one MI355X: a 1-rank NCCL(=RCCL) group still runs every collective the multi-GPU
layer issues (all_gather, gather, all_reduce, barrier) with device tensors."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

SCRIPT = r'''
import os, sys, torch
sys.path.insert(0, os.environ["TPZ_ROOT"])
from topaz_amd import parallel
rank, local_rank, world = parallel.init_from_env()          # backend nccl (RCCL)
assert torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl"
dev = torch.device("cuda", local_rank)
parallel.barrier(dev)
assert parallel.max_over_ranks(3.5, dev) == 3.5
s = [torch.tensor([2.5, -0.0, -1.25], device=dev), torch.zeros(0, device=dev)]
c = [torch.tensor([[1, 2], [3, 4], [5, 6]], dtype=torch.int32, device=dev), torch.zeros((0, 2), dtype=torch.int32, device=dev)]
out = parallel.gather_pick_tables([4, 9], s, c, dev)
assert sorted(out) == [4, 9] and torch.equal(out[4][0], s[0].cpu()) and torch.equal(out[4][1], c[0].cpu())
assert out[9][0].numel() == 0 and torch.signbit(out[4][0][1])      # -0.0 survives the int32 transport
torch.distributed.destroy_process_group()
print("RCCL_OK")
'''


def test_rccl_collectives_single_rank(gpu_ctx):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TPZ_ROOT=root, RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1',
               MASTER_PORT='29533', TOPAZ_AMD_FORCE_DIST='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, '-c', SCRIPT], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'RCCL_OK' in r.stdout, r.stdout + r.stderr


def test_bench_under_torchrun_single_gpu(gpu_ctx):
    """the driver launches N > 1 with torch.distributed.run; the same launcher must work for N = 1"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', '29534', os.path.join(root, 'bench.py'), '--gpus', '1', '--steps', '1', '--warmup', '1',
           '--size', '512', '--patch-size', '256', '--patch-padding', '64', '--no-cpu-baseline']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, TOPAZ_AMD_FORCE_DIST='1'))
    assert r.returncode == 0, r.stdout + r.stderr
    import json
    line = [l for l in r.stdout.splitlines() if l.startswith('{')][-1]
    d = json.loads(line)
    assert d['n_gpus'] == 1 and d['value'] > 0 and d['roofline']['achieved'] > 0


def _topaz(argv, env=None, stdin=None, timeout=900):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ if env is None else env)
    e['PYTHONPATH'] = root + os.pathsep + e.get('PYTHONPATH', '')
    return subprocess.run([sys.executable, '-m', 'topaz_amd'] + argv, env=e, capture_output=True, text=True, timeout=timeout,
                          input=stdin, cwd=root)


def _blob_mrcs(tmp_path, n, size=384):
    """n synthetic micrographs with dark blobs under the noise (real pick tables of different lengths per image)"""
    import numpy as np
    from topaz_amd.utils.image import save_image
    paths = []
    for i in range(n):
        rs = np.random.RandomState(500 + i)
        x = rs.randn(size, size + 16 * (i % 3)).astype(np.float32)
        yy, xx = np.mgrid[0:x.shape[0], 0:x.shape[1]].astype(np.float32)
        for cy, cx in rs.randint(20, size - 20, size=(10 + 7 * i, 2)):
            x -= 2.5 * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * 6.0 ** 2)).astype(np.float32)
        p = str(tmp_path / f'mic_{i:02d}.mrc')
        save_image(x, p)
        paths.append(p)
    return paths


def test_two_ranks_share_the_gpu_extract_tsv_is_byte_identical(gpu_ctx, tmp_path):
    """BASELINE config 4's job path without an 8-GPU node: `topaz extract --gpus 2` starts two rank
    processes; here both drive GPU 0 (TOPAZ_AMD_SHARE_GPU=1) with the HIP path for compute and exchange their REAL pick tables
    over gloo (RCCL refuses two ranks on one device; the sharding i = rank (mod N), the device-side packing of the tables, the
    size all_gather + the one gather and rank 0's assembly of the TSV in input order do not care).  The gathered TSV equals the
    single-process run byte for byte; so does the stdin form of the input list; exit status 0."""
    mics = _blob_mrcs(tmp_path, 5)
    one, two, three = str(tmp_path / 'one.txt'), str(tmp_path / 'two.txt'), str(tmp_path / 'three.txt')
    base = ['extract', '-m', 'resnet8_u32', '-r', '8', '-t', '-6']
    r = _topaz(base + ['-o', one] + mics)
    assert r.returncode == 0, r.stderr[-2000:]
    env = dict(os.environ, TOPAZ_AMD_SHARE_GPU='1', TOPAZ_AMD_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = _topaz(base + ['--gpus', '2', '-o', two] + mics, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    a, b = open(one, 'rb').read(), open(two, 'rb').read()
    assert a.count(b'\n') > 40                                   # (real picks, every image)
    for i in range(5):
        assert f'mic_{i:02d}\t'.encode() in a
    assert a == b
    r = _topaz(base + ['--gpus', '2', '-o', three], env=env, stdin='\n'.join(mics) + '\n')
    assert r.returncode == 0, r.stderr[-2000:]
    assert open(three, 'rb').read() == a


def test_two_ranks_share_the_gpu_denoise_and_denoise3d_exit_status(gpu_ctx, tmp_path):
    """`topaz denoise --gpus 2` / `topaz denoise3d --gpus 2`: the commands return their output / input lists to a Python caller,
    which must not become the exit status (round-4 advisor finding: every rank exited 1 and the launcher killed its peers).
    Two ranks on GPU 0 (gloo): every output file exists and equals the single-process run's; a tomogram split by TILES over
    the ranks (fewer volumes than ranks: tpz_denoise_3d_shard + one reduce) equals the single-process volume bit for bit."""
    mics = _blob_mrcs(tmp_path, 3, size=256)
    env = dict(os.environ, TOPAZ_AMD_SHARE_GPU='1', TOPAZ_AMD_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    d1, d2 = str(tmp_path / 'den1'), str(tmp_path / 'den2')
    args = ['denoise', '-m', 'unet-v0.2.1', '-s', '128', '-p', '32']
    r = _topaz(args + ['-o', d1] + mics)
    assert r.returncode == 0, r.stderr[-2000:]
    r = _topaz(args + ['--gpus', '2', '-o', d2] + mics, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    for p in mics:
        n = os.path.basename(p)
        assert open(os.path.join(d1, n), 'rb').read() == open(os.path.join(d2, n), 'rb').read()
    # one tomogram, two ranks: tiles dealt to the ranks, partial volumes summed onto rank 0
    from conftest import GOLDEN
    cli = os.path.join(GOLDEN, 'cli')
    tp = str(tmp_path / 'tomo.mrc')
    import shutil
    shutil.copy(os.path.join(cli, 'tomo.mrc'), tp)
    t1, t2 = str(tmp_path / 't1'), str(tmp_path / 't2')
    a3 = ['denoise3d', '-m', os.path.join(cli, 'unet3d_nf8_state.sav'), '--base-kernel-width', '7', '-s', '32', '-p', '16', '-d', '0']
    r = _topaz(a3 + ['-o', t1, tp])
    assert r.returncode == 0, r.stderr[-2000:]
    r = _topaz(a3 + ['--gpus', '2', '-o', t2, tp], env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert open(os.path.join(t1, 'tomo.mrc'), 'rb').read() == open(os.path.join(t2, 'tomo.mrc'), 'rb').read()


def test_bench_two_ranks_share_the_gpu(gpu_ctx):
    """`python bench.py --gpus 2` as the driver's scaling run would execute it -- self-launched ranks, barrier + max-over-ranks
    timing, the gather of REAL pick tables, one JSON line from rank 0 -- rehearsed on one GPU: both ranks drive GPU 0
    (TOPAZ_AMD_SHARE_GPU=1), the collectives run over gloo on host tensors.  Not a measurement (two ranks share one device);
    what is checked is the multi-rank code path of the bench itself."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TOPAZ_AMD_SHARE_GPU='1', TOPAZ_AMD_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--size', '1024',
           '--patch-size', '512', '--patch-padding', '128', '--no-cpu-baseline']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1                                       # rank 0 alone prints
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['rccl_world'] == 2 and d['config']['images_total'] == 4 and d['value'] > 0
    assert len(d['rank_ms_per_step']['all']) == 2 and d['config']['picks_per_image'] > 10
    assert d['roofline']['achieved'] > 0 and 'configs' not in d and 'cpu_baseline' not in d


def test_bench_line_carries_parity_and_fails_on_a_broken_one(gpu_ctx):
    """bench.py's own parity check: the default line carries `parity` -- the HIP path against the oracle on the
    arrays of the cpu_baseline leg -- and `energy`, `leg_seconds`; the process exits 0 only when the bars hold.  The failing side is
    checked on the function itself: a tampered oracle array makes `ok` false (bench.py then exits 3)."""
    import argparse
    import json
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--steps', '1', '--warmup', '1', '--size', '768', '--patch-size', '384',
           '--patch-padding', '96', '--cpu-sample', '512', '--cpu-score-sample', '512', '--no-configs']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    p = d['parity']
    assert p['ok'] and p['picks_equal'] and p['denoise_max_abs'] <= 1e-4 and p['logit_max_abs'] <= 1e-4 and p['picks'] > 50
    assert d['cpu_baseline']['value'] > 0 and 'cpu_baseline' in d['leg_seconds'] and d['host_placement']['host_cpus_before_pinning']
    assert 'cli_inclusive' not in d and 'pcie_inclusive' not in d                    # secondary legs: --extras only
    # the failing side
    sys.path.insert(0, root)
    import torch
    import bench
    from oracle import nms as onms
    from oracle import scoring as oscoring
    models = bench.build_models('extract')
    x = np.random.RandomState(5).randn(256, 256).astype(np.float32)
    logit = oscoring.score('resnet8', models['score'][1], x)
    args = argparse.Namespace(radius=14, threshold=-6.0)
    keep = {'score_in': x, 'logit': logit, 'picks': onms.nms2d(logit, 14, -6.0)}
    good = bench.parity_vs_oracle(models, keep, args, torch.device('cuda', 0))
    assert good['ok'] and good['picks_equal'] and good['logit_max_abs'] <= 1e-4
    bad = dict(keep, logit=logit + np.float32(3e-4))
    out = bench.parity_vs_oracle(models, bad, args, torch.device('cuda', 0))
    assert not out['ok'] and out['logit_max_abs'] > 1e-4
