#!/usr/bin/env python
"""Rehearsal of the multi-rank bench on ONE GPU: `python bench.py --gpus N` with N = 1, 2, 4, 8 rank
processes that all drive GPU 0 (TOPAZ_AMD_SHARE_GPU=1, collectives over gloo on host tensors).  Every rank does the same
work as in the real job -- its own launch thread, pinned staging, NMS host reads, the pick-table gather -- only the device is
shared, so the AGGREGATE throughput of N ranks should equal the 1-rank value: whatever is lost is the host path serialising
(launch threads contending, the gather, Python start-up skew), which is what would break the scaling on a real 8-GPU node
first.  Not a measurement of multi-GPU throughput: there is one GPU.

    python tools/rehearse_ranks.py [--size 2048] [--steps 6] > profiles/r06_rank_rehearsal.json
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(n: int, args) -> dict:
    env = dict(os.environ, TOPAZ_AMD_SHARE_GPU='1', TOPAZ_AMD_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n), '--steps', str(args.steps), '--warmup', '2',
           '--size', str(args.size), '--patch-size', str(args.patch_size), '--patch-padding', str(args.patch_padding),
           '--no-cpu-baseline', '--no-extras', '--no-configs', '--no-kernel-timing']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    if r.returncode != 0 or len(lines) != 1:
        return {'ranks': n, 'error': (r.stdout[-800:] + r.stderr[-1200:])}
    d = json.loads(lines[0])
    trace = [l for l in r.stderr.splitlines() if l.startswith('[topaz_amd gather]')]      # (TOPAZ_AMD_TRACE_GATHER=1)
    return {'ranks': n, **({'gather_trace': trace[-2 * n:]} if trace else {}), 'micrographs_per_s_all_ranks': d['value'], 'ms_per_step_per_rank': d['ms_per_step'],
            'rank_ms_per_step': d['rank_ms_per_step'], 'rank_host_cpu_ms_per_step': d['rank_host_cpu_ms_per_step'],
            'gather_ms': d['gather_ms'], 'images_total': d['config']['images_total'], 'rccl_world': d['rccl_world'],
            'picks_per_image': d['config']['picks_per_image']}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', type=int, default=2048)
    ap.add_argument('--patch-size', type=int, default=512)
    ap.add_argument('--patch-padding', type=int, default=250)
    ap.add_argument('--steps', type=int, default=6)
    ap.add_argument('--ranks', type=int, nargs='*', default=[1, 2, 4, 8])
    args = ap.parse_args()
    rows = [run(n, args) for n in args.ranks]
    base = next((r for r in rows if r.get('ranks') == 1 and 'error' not in r), None)
    for r in rows:
        if base and 'error' not in r:
            r['aggregate_vs_one_rank'] = r['micrographs_per_s_all_ranks'] / base['micrographs_per_s_all_ranks']
            # ... and of the compute phase alone (each rank's own steps / its own compute time, summed): the exchange step here runs
            # over gloo / TCP between processes that share one GPU and its NUMA node -- on the real node it is one RCCL gather
            r['compute_aggregate_vs_one_rank'] = (sum(1e3 / v for v in r['rank_ms_per_step']['all']) /
                                                  sum(1e3 / v for v in base['rank_ms_per_step']['all']))
    print(json.dumps({
        'what': f'bench.py --gpus N with every rank on GPU 0 (TOPAZ_AMD_SHARE_GPU=1, gloo), {args.size}^2 micrographs, '
                f'-s {args.patch_size} -p {args.patch_padding}, {args.steps} steps per rank; aggregate_vs_one_rank = throughput summed '
                'over the N rank processes / the 1-rank value (1.0 = the host path adds nothing when ranks multiply)',
        'note': 'one GPU: this rehearses the HOST side of an N-rank job, it is not a multi-GPU measurement',
        'rows': rows}, indent=1))


if __name__ == '__main__':
    main()
