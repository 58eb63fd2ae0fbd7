"""gloo tests (world 2 and world 8) of the multi-GPU layer: sharding, the pick-table gather, the launcher, bench.py's ranks."""
import os
import socket
import subprocess
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _tables(image_id):
    rs = np.random.RandomState(100 + image_id)
    n = int(rs.randint(0, 40))
    s = torch.from_numpy(np.sort(rs.randn(n).astype(np.float32))[::-1].copy())
    c = torch.from_numpy(rs.randint(0, 4096, size=(n, 2)).astype(np.int32))
    return s, c


def _worker(rank, world, port, n_images, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from topaz_amd import parallel
    r, lr, w = parallel.init_from_env('gloo')
    assert (r, w) == (rank, world)
    mine = parallel.shard_indices(n_images, rank, world)
    tabs = [_tables(i) for i in mine]
    dev = torch.device('cpu')
    parallel.barrier(dev)
    t = parallel.max_over_ranks(float(rank + 1), dev)
    assert t == float(world)
    out = parallel.gather_pick_tables(mine, [a for a, _ in tabs], [b for _, b in tabs], dev)
    if rank == 0:
        ok = sorted(out) == list(range(n_images))
        for i in range(n_images):
            s, c = _tables(i)
            ok = ok and torch.equal(out[i][0], s) and torch.equal(out[i][1], c)
        q.put(ok)
    else:
        assert out is None
    torch.distributed.destroy_process_group()


def test_shard_indices_cover_everything():
    from topaz_amd.parallel import shard_indices
    for n in (0, 1, 7, 256):
        for w in (1, 2, 8):
            got = sorted(i for r in range(w) for i in shard_indices(n, r, w))
            assert got == list(range(n))
    assert shard_indices(256, 3, 8)[:3] == [3, 11, 19] and len(shard_indices(256, 3, 8)) == 32


def test_gather_pick_tables_world2_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 7, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_gather_pick_tables_world8_gloo():
    """BASELINE config 4's shape: eight ranks, images dealt i = rank (mod 8), ragged tables (some empty), one gather"""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 8, port, 37, q)) for r in range(8)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_gather_single_process():
    from topaz_amd.parallel import gather_pick_tables
    s, c = _tables(3)
    out = gather_pick_tables([3], [s], [c], torch.device('cpu'))
    assert torch.equal(out[3][0], s) and torch.equal(out[3][1], c)


def test_bench_self_launches_its_ranks_dry_run():
    """plain `python bench.py --gpus 2` (no torchrun, WORLD_SIZE unset) starts its own two rank processes and rank 0
    prints ONE JSON line with n_gpus = 2; --dry-run replaces the hot path (no GPU here) by fabricated pick tables that
    go through the same barriers / max-over-ranks / gather over gloo"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--dry-run'],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['steps'] == 3 and out['images_gathered'] == 6 and out['dry_run'] is True
    assert out['rccl_world'] == 2 and out['scaling'] == 'weak'


def test_bench_eight_ranks_dry_run_weak_and_strong():
    """`python bench.py --gpus 8`: eight self-launched ranks (gloo here, RCCL on the node) -- the collective sees all eight
    (`rccl_world`), the weak job gathers 8 x steps tables, the strong job (BASELINE config 4: a FIXED set of micrographs dealt
    i = rank mod 8) gathers exactly --images of them, ragged shares included"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    for extra, n_img in ((['--steps', '2'], 16), (['--scaling', 'strong', '--images', '21'], 21)):
        r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '8', '--dry-run'] + extra,
                           capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
        assert len(lines) == 1, r.stdout
        out = json.loads(lines[0])
        assert out['n_gpus'] == 8 and out['rccl_world'] == 8 and out['images_gathered'] == n_img, out
        assert out['rank_ms_per_step']['max'] >= out['rank_ms_per_step']['min'] > 0


def test_launch_local_ranks_propagates_failure():
    import sys
    from topaz_amd.parallel import launch_local_ranks
    code = ('import os, sys, time\n'
            'r = int(os.environ["RANK"]); assert os.environ["WORLD_SIZE"] == "2" and os.environ["LOCAL_RANK"] == str(r)\n'
            'assert os.environ["MASTER_ADDR"] == "127.0.0.1"\n'
            'if r == 1: sys.exit(3)\n'
            'time.sleep(60)\n')
    import time
    t0 = time.time()
    assert launch_local_ranks(2, [sys.executable, '-c', code]) == 3        # rank 0 is terminated, not waited for
    assert time.time() - t0 < 30
    assert launch_local_ranks(2, [sys.executable, '-c', 'import os; assert "RANK" in os.environ']) == 0
    # ranks are started without a stdin to fight over
    assert launch_local_ranks(2, [sys.executable, '-c', 'import sys; sys.exit(0 if sys.stdin.read() == "" else 5)']) == 0


def test_launcher_reads_stdin_once_for_all_ranks(monkeypatch):
    """`topaz extract --gpus N` with the micrograph list on stdin: the launcher reads stdin ONCE and hands every rank the whole
    list OUT OF BAND (a file named in TOPAZ_AMD_INPUT_LIST); rank processes get no stdin and an unchanged command line.  (N ranks
    sharing one pipe would each read a part of it, take that part for the whole list and shard it again; an @file token would be
    re-parsed by the ranks: names starting with '-' or '@', or a trailing variadic option, would be misread.)"""
    import io
    import sys
    from topaz_amd import main as tmain
    from topaz_amd import parallel
    seen = {}

    def fake_launch(n, cmd, env=None, **kw):
        seen['n'] = n
        seen['cmd'] = list(cmd)
        seen['list'] = env['TOPAZ_AMD_INPUT_LIST']
        seen['names'] = open(seen['list']).read().split('\n')[:-1]
        return 0

    monkeypatch.setattr(parallel, 'launch_local_ranks', fake_launch)
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    names = [f'/data/mic_{i:03d}.mrc' for i in range(35)] + ['-odd name.mrc', '@at.mrc']
    monkeypatch.setattr(sys, 'stdin', io.StringIO('\n'.join(names) + '\n\n'))
    argv = ['extract', '-m', 'resnet8_u32', '-r', '8', '--gpus', '4']
    assert tmain.main(argv) == 0
    assert seen['n'] == 4 and seen['names'] == names
    assert seen['cmd'][-len(argv):] == argv                   # nothing appended to the ranks' command line
    assert not os.path.exists(seen['list'])                   # the list file is removed afterwards
    # a rank (no paths on its command line) takes the list from that file, names untouched
    from topaz_amd import extract as ext
    lf = os.path.join(os.path.dirname(__file__), '_tmp_list.txt')
    open(lf, 'w').write('\n'.join(names) + '\n')
    monkeypatch.setenv('TOPAZ_AMD_INPUT_LIST', lf)
    monkeypatch.setenv('WORLD_SIZE', '1')                     # only a rank process of a launcher honours the variable ...
    got = {}

    def fake_score(model, paths, **kw):
        got['paths'] = list(paths)
        raise KeyboardInterrupt                               # (stop before any GPU work)

    monkeypatch.setattr(ext, 'score_images', fake_score)
    monkeypatch.setattr(ext.parallel, 'init_from_env', lambda *a, **k: (0, 0, 1), raising=False)
    try:
        ext.extract_particles([], 'resnet8_u32', 0, 1, -6.0, 8, 0, None, 5, 100, 5, -1, 0, False, None, False, '', 'coord', 1.0, 1.0)
    except KeyboardInterrupt:
        pass
    finally:
        os.unlink(lf)
    assert got.get('paths') == names
    assert 'TOPAZ_AMD_INPUT_LIST' not in os.environ           # ... once (popped: nothing stale is left behind)
    # outside a launcher an exported / stale variable is ignored: stdin is the list (topaz/extract.py:270)
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    monkeypatch.setenv('TOPAZ_AMD_INPUT_LIST', '/nonexistent/stale_list.txt')
    monkeypatch.setattr(sys, 'stdin', io.StringIO('a.mrc\nb.mrc\n'))
    try:
        ext.extract_particles([], 'resnet8_u32', 0, 1, -6.0, 8, 0, None, 5, 100, 5, -1, 0, False, None, False, '', 'coord', 1.0, 1.0)
    except KeyboardInterrupt:
        pass
    assert got.get('paths') == ['a.mrc', 'b.mrc']


def test_cli_exit_status_ignores_command_return_values(monkeypatch):
    """`topaz denoise` / `denoise3d` return their output / input lists to a Python caller (topaz/denoise.py:463-487, 548);
    the console script does `raise SystemExit(main())`, so main() must return 0 after a successful command -- a list there
    prints itself and exits 1, and under `--gpus N` the launcher then kills the ranks still working.  Like topaz/main.py:148."""
    from topaz_amd import main as tmain
    from topaz_amd.commands import denoise, denoise3d
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    monkeypatch.setattr(denoise, 'main', lambda args: ['out/a.mrc', 'out/b.mrc'])
    monkeypatch.setattr(denoise3d, 'main', lambda args: [])
    assert tmain.main(['denoise', 'a.mrc', 'b.mrc']) == 0
    assert tmain.main(['denoise3d', 'a.mrc']) == 0
    # and through the module entry point: exit status 0, nothing but the command's own output
    code = ('import sys; from topaz_amd.commands import denoise; denoise.main = lambda a: ["x.mrc"]; '
            'sys.argv = ["topaz", "denoise", "x.mrc"]; import runpy; runpy.run_module("topaz_amd", run_name="__main__")')
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300,
                       cwd=os.path.join(os.path.dirname(__file__), '..'))
    assert r.returncode == 0, r.stderr[-400:]
    assert 'x.mrc' not in r.stderr


def _sum_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from topaz_amd import parallel
    parallel.init_from_env('gloo')
    # each rank owns the tiles t = rank (mod world) of a 4 x 4 x 4 tile grid of 8^3 tiles
    vol = torch.zeros(32, 32, 32)
    t = 0
    for i in range(0, 32, 8):
        for j in range(0, 32, 8):
            for k in range(0, 32, 8):
                if t % world == rank:
                    vol[i:i + 8, j:j + 8, k:k + 8] = float(t + 1)
                t += 1
    out = parallel.sum_to_root(vol)
    if rank == 0:
        want = torch.arange(1, 65, dtype=torch.float32).reshape(4, 4, 4).repeat_interleave(8, 0).repeat_interleave(8, 1).repeat_interleave(8, 2)
        q.put(bool(torch.equal(out, want)))
    else:
        assert out is None
    torch.distributed.destroy_process_group()


def test_sum_to_root_assembles_tile_sharded_volume_world2_gloo():
    """the exchange step of tile-sharded tomogram denoising: every voxel is non-zero on exactly one rank, one reduce onto
    rank 0 assembles the volume"""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sum_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_rank_cpu_sets_follow_the_gpu_numa_nodes(tmp_path):
    """launch_local_ranks pins rank r to CPUs of GPU r's NUMA node (sysfs: /sys/class/drm/cardN/device/numa_node and the
    node's cpulist), GPUs of one node sharing it in equal slices; unreadable topology -> equal slices of the allowed CPUs.
    A rank really starts with that affinity."""
    import sys
    from topaz_amd.parallel import cpu_sets_for_ranks, launch_local_ranks
    root = tmp_path / 'sys'
    for card, node in enumerate([0, 0, 1, 1]):
        d = root / 'class' / 'drm' / f'card{card}' / 'device'
        d.mkdir(parents=True)
        (d / 'numa_node').write_text(f'{node}\n')
    (root / 'class' / 'drm' / 'card0-DP-1').mkdir()               # a connector entry: not a device
    for node, cpus in ((0, '0-7,16-23'), (1, '8-15,24-31')):
        d = root / 'devices' / 'system' / 'node' / f'node{node}'
        d.mkdir(parents=True)
        (d / 'cpulist').write_text(cpus + '\n')
    sets = cpu_sets_for_ranks(4, sysfs=str(root), allowed=range(32))
    assert sets == [[0, 1, 2, 3, 4, 5, 6, 7], [16, 17, 18, 19, 20, 21, 22, 23], [8, 9, 10, 11, 12, 13, 14, 15],
                    [24, 25, 26, 27, 28, 29, 30, 31]]
    # restricted affinity mask of the launcher is respected
    assert cpu_sets_for_ranks(2, sysfs=str(root), allowed=range(0, 8)) == [[0, 1, 2, 3], [4, 5, 6, 7]]
    # more ranks than GPUs listed, or no topology at all: equal contiguous slices
    assert cpu_sets_for_ranks(8, sysfs=str(root), allowed=range(16)) == [[2 * i, 2 * i + 1] for i in range(8)]
    assert cpu_sets_for_ranks(2, sysfs=str(tmp_path / 'nothing'), allowed=range(6)) == [[0, 1, 2], [3, 4, 5]]
    # numa_node -1 (single-socket hosts report it): fallback as well
    (root / 'class' / 'drm' / 'card1' / 'device' / 'numa_node').write_text('-1\n')
    assert cpu_sets_for_ranks(2, sysfs=str(root), allowed=range(4)) == [[0, 1], [2, 3]]
    # under a *_VISIBLE_DEVICES subset HIP device r is not DRM card r: no guessing from the card order (the rank resolves its
    # own GPU through HIP's PCI address instead: explicit `nodes`)
    (root / 'class' / 'drm' / 'card1' / 'device' / 'numa_node').write_text('0\n')
    monkey_env = dict(os.environ)
    try:
        os.environ['ROCR_VISIBLE_DEVICES'] = '2,3'
        assert cpu_sets_for_ranks(2, sysfs=str(root), allowed=range(32)) == [list(range(16)), list(range(16, 32))]
        assert cpu_sets_for_ranks(2, sysfs=str(root), allowed=range(32), nodes=[1, 1]) == [
            [8, 9, 10, 11, 12, 13, 14, 15], [24, 25, 26, 27, 28, 29, 30, 31]]
    finally:
        os.environ.clear()
        os.environ.update(monkey_env)
    # a launched rank pins ITSELF (every thread it already has) to the CPUs the launcher announced to it
    code = ('import os, sys, threading, time\n'
            f'sys.path.insert(0, {ROOT!r})\n'
            'stop = threading.Event()\n'
            'th = threading.Thread(target=stop.wait, daemon=True); th.start()\n'
            'from topaz_amd.parallel import pin_this_rank\n'
            'want = sorted(int(c) for c in os.environ["TOPAZ_AMD_RANK_CPUS"].split(","))\n'
            'got = pin_this_rank(int(os.environ["LOCAL_RANK"]), int(os.environ["LOCAL_WORLD_SIZE"]))\n'
            'ok = sorted(got or []) == want and all(sorted(os.sched_getaffinity(int(t))) == want for t in os.listdir("/proc/self/task"))\n'
            'stop.set()\n'
            'sys.exit(0 if ok and len(os.listdir("/proc/self/task")) >= 1 else 7)\n')
    assert launch_local_ranks(2, [sys.executable, '-c', code]) == 0
