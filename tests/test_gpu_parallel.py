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
