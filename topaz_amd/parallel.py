"""Multi-GPU layer: micrographs shard embarrassingly (one image per rank at a time, rank r takes
images i = r (mod world)); the only exchange step is the gather of the per-image pick tables to
rank 0 for the single-TSV output mode of `topaz extract` (topaz/extract.py:321-354).

One process per GPU; torch.distributed backend 'nccl' (= RCCL over xGMI on ROCm) on the GPU box,
'gloo' in the CPU tests.  The reference has no distributed code; this is new.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """(rank, local_rank, world).  Single process when WORLD_SIZE is unset or 1."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1:
        pin_this_rank(local_rank, int(os.environ.get('LOCAL_WORLD_SIZE', world)))
    if (world > 1 or os.environ.get('TOPAZ_AMD_FORCE_DIST') == '1') and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local_rank, world


def under_launcher() -> bool:
    """True inside a rank process (torchrun or launch_local_ranks set WORLD_SIZE)"""
    return 'WORLD_SIZE' in os.environ


def free_port() -> int:
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _parse_cpulist(text: str) -> List[int]:
    out: List[int] = []
    for part in text.strip().split(','):
        if not part:
            continue
        lo, _, hi = part.partition('-')
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def cpu_sets_for_ranks(n: int, sysfs: str = '/sys', allowed: Optional[Sequence[int]] = None) -> List[List[int]]:
    """Host placement of n rank processes on one node: rank r gets CPUs of the NUMA node GPU r hangs off (the amdgpu render
    devices of /sys/class/drm in card order, their device/numa_node, that node's cpulist), the GPUs of one node sharing its CPUs
    in equal contiguous slices -- each rank's launch thread (~700 kernel launches per micrograph) and reader thread then stay
    next to their GPU's PCIe root and off the other ranks' cores.  Fallback when the topology cannot be read (no sysfs entry,
    numa_node -1, fewer GPUs listed than ranks): the allowed CPUs cut into n equal contiguous slices.  Pure function of the
    sysfs tree (tests feed it a fake one)."""
    import glob
    if allowed is None:
        try:
            allowed = sorted(os.sched_getaffinity(0))
        except (AttributeError, OSError):
            allowed = list(range(os.cpu_count() or 1))
    allowed = list(allowed)

    def even_slices(cpus: Sequence[int], k: int) -> List[List[int]]:
        cpus = list(cpus)
        if len(cpus) < k:
            return [cpus for _ in range(k)]
        return [cpus[i * len(cpus) // k:(i + 1) * len(cpus) // k] for i in range(k)]

    nodes: List[int] = []
    try:
        cards = sorted(glob.glob(os.path.join(sysfs, 'class/drm/card[0-9]*')),
                       key=lambda p: int(''.join(ch for ch in os.path.basename(p) if ch.isdigit())))
        for c in cards:
            if '-' in os.path.basename(c):                       # connectors (card0-DP-1), not devices
                continue
            drv = os.path.join(c, 'device/driver')
            if os.path.exists(drv) and os.path.basename(os.path.realpath(drv)) != 'amdgpu':
                continue
            nodes.append(int(open(os.path.join(c, 'device/numa_node')).read().strip()))
    except (OSError, ValueError):
        nodes = []
    if len(nodes) < n or any(v < 0 for v in nodes[:n]):
        return even_slices(allowed, n)
    nodes = nodes[:n]
    out: List[List[int]] = [[] for _ in range(n)]
    for node in sorted(set(nodes)):
        try:
            cpus = [c for c in _parse_cpulist(open(os.path.join(sysfs, f'devices/system/node/node{node}/cpulist')).read())
                    if c in set(allowed)]
        except (OSError, ValueError):
            return even_slices(allowed, n)
        ranks = [r for r in range(n) if nodes[r] == node]
        if not cpus:
            return even_slices(allowed, n)
        for r, sl in zip(ranks, even_slices(cpus, len(ranks))):
            out[r] = sl
    return out


def pin_this_rank(local_rank: int, local_world: int) -> Optional[List[int]]:
    """host placement of a rank that somebody else started (torchrun): the same CPU set launch_local_ranks would have given
    it.  No-op when our own launcher pinned it already, when TOPAZ_AMD_NO_AFFINITY=1, or where affinity cannot be set."""
    if os.environ.get('TOPAZ_AMD_RANK_CPUS') or os.environ.get('TOPAZ_AMD_NO_AFFINITY') == '1' or not hasattr(os, 'sched_setaffinity'):
        return None
    try:
        sets = cpu_sets_for_ranks(local_world)
        if 0 <= local_rank < len(sets) and sets[local_rank]:
            os.sched_setaffinity(0, set(sets[local_rank]))
            os.environ['TOPAZ_AMD_RANK_CPUS'] = ','.join(map(str, sets[local_rank]))
            return sets[local_rank]
    except OSError:
        pass
    return None


def launch_local_ranks(n: int, argv: Sequence[str], env: Optional[dict] = None, timeout: Optional[float] = None) -> int:
    """One process per GPU without an external launcher: start `n` copies of `argv` (a full command line, e.g.
    [sys.executable, 'bench.py', '--gpus', '8']) with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set,
    rank r on GPU r, all on 127.0.0.1.  stdout / stderr are inherited, so rank 0's output is the job's output.
    Returns the largest exit code; when a rank fails the others are terminated (a hung collective would otherwise
    wait for its peer forever).  The counterpart of the reference's `-d -2` "use every GPU" (commands/denoise3d.py:
    102-103,117-118 -- DataParallel threads there, processes over RCCL here)."""
    import subprocess
    import time
    base = dict(os.environ if env is None else env)
    base.update(WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(free_port()))
    base.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    procs = []
    # host placement: rank r on the CPUs next to GPU r (cpu_sets_for_ranks; TOPAZ_AMD_NO_AFFINITY=1 leaves the ranks unpinned)
    cpu_sets = None if base.get('TOPAZ_AMD_NO_AFFINITY') == '1' else cpu_sets_for_ranks(n)
    for r in range(n):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        pin = None
        if cpu_sets and cpu_sets[r] and hasattr(os, 'sched_setaffinity'):
            e['TOPAZ_AMD_RANK_CPUS'] = ','.join(map(str, cpu_sets[r]))
            pin = (lambda cpus: (lambda: os.sched_setaffinity(0, cpus)))(set(cpu_sets[r]))
        # (stdin is not shared: N readers of one pipe would each see a part of it -- the launcher resolves stdin input itself)
        procs.append(subprocess.Popen(list(argv), env=e, stdin=subprocess.DEVNULL, preexec_fn=pin))
    t0 = time.time()
    codes: List[Optional[int]] = [None] * n
    try:
        while any(c is None for c in codes):
            for r, p in enumerate(procs):
                if codes[r] is None:
                    codes[r] = p.poll()
            failed = [c for c in codes if c not in (None, 0)]
            if failed or (timeout is not None and time.time() - t0 > timeout):
                for r, p in enumerate(procs):
                    if codes[r] is None:
                        p.terminate()
                for r, p in enumerate(procs):
                    if codes[r] is None:
                        try:
                            codes[r] = p.wait(10)
                        except subprocess.TimeoutExpired:
                            p.kill()
                            codes[r] = p.wait()
                return max([abs(c) for c in failed] or [124])
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return max(abs(c) for c in codes)


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """items processed by `rank`: i = rank (mod world), in input order"""
    return list(range(rank, n_items, world))


def barrier(device: Optional[torch.device] = None) -> None:
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if device is not None and device.type == 'cuda':
            dist.barrier(device_ids=[device.index])
        else:
            dist.barrier()


def max_over_ranks(value: float, device: torch.device) -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device: torch.device) -> float:
    """all_reduce(SUM) of one number; `value` itself without a process group"""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_scalars(value: float, device: torch.device) -> List[float]:
    """every rank's number, on every rank (one all_gather)"""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [value]
    t = torch.tensor([value], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def sum_to_root(t: torch.Tensor, dst: int = 0) -> Optional[torch.Tensor]:
    """element-wise sum of every rank's tensor onto rank `dst` (one reduce; RCCL on device tensors, gloo on host ones).
    Used to assemble a tomogram whose tiles were denoised by different ranks: each voxel is non-zero on exactly one rank, so
    the sum is exact.  Returns the tensor on dst, None elsewhere; a single process returns its tensor unchanged."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return t
    dist.reduce(t, dst=dst, op=dist.ReduceOp.SUM)
    return t if dist.get_rank() == dst else None


def gather_pick_tables(image_ids: Sequence[int], scores: Sequence[torch.Tensor], coords: Sequence[torch.Tensor],
                       device: torch.device, dst: int = 0):
    """Gather per-image pick tables to rank `dst`.

    Every rank passes the images it owns: ids, scores[i] (fp32 [n_i]) and coords[i] (int32 [n_i, d]).
    Exchange: one all_gather of (image count, row count) per rank, then one gather of the
    concatenated (id, n) headers and one of the (x, y[, z], score-bits) rows, padded to the largest
    rank.  Rows are a few hundred KB per micrograph, so the step is latency-bound; each peer writes to
    the root over its own xGMI link.  Returns on dst a dict id -> (scores, coords) (CPU tensors), on
    the other ranks None.
    """
    d = coords[0].shape[1] if len(coords) else 2
    if not (dist.is_available() and dist.is_initialized()):
        return {int(i): (s.cpu(), c.cpu()) for i, s, c in zip(image_ids, scores, coords)}
    world, rank = dist.get_world_size(), dist.get_rank()      # (a 1-rank group takes the collective path too)
    n_img = len(image_ids)
    n_rows = int(sum(int(s.numel()) for s in scores))
    meta = torch.tensor([n_img, n_rows, d], dtype=torch.int64, device=device)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)
    max_img = max(int(m[0]) for m in metas)
    max_rows = max(int(m[1]) for m in metas)
    d = max(int(m[2]) for m in metas)
    head = torch.zeros((max(max_img, 1), 2), dtype=torch.int64, device=device)
    for k, (i, s) in enumerate(zip(image_ids, scores)):
        head[k, 0], head[k, 1] = int(i), int(s.numel())
    rows = torch.zeros((max(max_rows, 1), d + 1), dtype=torch.int32, device=device)
    if n_rows:
        cat_c = torch.cat([c.to(device=device, dtype=torch.int32).reshape(-1, d) for c in coords], 0)
        cat_s = torch.cat([s.to(device=device, dtype=torch.float32).reshape(-1) for s in scores], 0)
        rows[:n_rows, :d] = cat_c
        rows[:n_rows, d] = cat_s.view(torch.int32)          # bit-exact transport of the fp32 scores
    heads = [torch.zeros_like(head) for _ in range(world)] if rank == dst else None
    rowss = [torch.zeros_like(rows) for _ in range(world)] if rank == dst else None
    dist.gather(head, heads, dst=dst)
    dist.gather(rows, rowss, dst=dst)
    if rank != dst:
        return None
    out = {}
    for r in range(world):
        h, rw = heads[r].cpu(), rowss[r].cpu()
        off = 0
        for k in range(int(metas[r][0])):
            iid, n = int(h[k, 0]), int(h[k, 1])
            blk = rw[off:off + n]
            out[iid] = (blk[:, d].contiguous().view(torch.float32).clone(), blk[:, :d].clone())
            off += n
    return out
