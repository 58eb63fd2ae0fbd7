#!/usr/bin/env python
"""HBM traffic of the benchmark's dominant kernel from rocprofv3 PMC passes -> profiles/pmc_dominant.json.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE  -d gpurun_out/pmc_fetch -o p --output-format csv -- python bench.py --steps 2 ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE  -d gpurun_out/pmc_write -o p --output-format csv -- python bench.py --steps 2 ...
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write [--size 4096] [--workload pipeline]

FETCH_SIZE and WRITE_SIZE need separate passes on gfx950 (3 + 2 of the 4 TCC slots).  Both are reported in KiB per
dispatch.  Correction (MI355X_MICROARCH.md, HBM section): on gfx950 FETCH_SIZE tallies the 128-byte requests of a wide
coalesced stream (16 B per lane -- what the LDS-DMA input stream of conv_split_kernel is) at 64 bytes, i.e. reports half
of the bytes: it is doubled.  WRITE_SIZE is taken as reported (uncalibrated in the guide).  bench.py reports the result as
roofline.traffic when the kernel name, image size and workload match.
"""
import argparse
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bench_name(rocprof_name: str):
    """tpz::conv_split_kernel<tpz::SplitCfg<5, 4, 128, 16, 32, 2, 8, 5>, 3, 0>(...) -> the registry name bench.py prints"""
    m = re.search(r'conv_split_kernel<tpz::SplitCfg<([\d, ]+)>, (\d+), (\d+)(?:, \d+)?>', rocprof_name)
    if not m:
        return None
    k, d, mt, th, tw, cc, w, kx = [int(v) for v in m.group(1).split(',')[:8]]
    return f'conv_split_kernel<K={k}x{kx},D={d},MT={mt},TH={th},TW={tw},CC={cc},W={w},EPI={m.group(2)}>'


def per_dispatch(directory: str, counter: str):
    vals = collections.defaultdict(list)
    for f in glob.glob(os.path.join(directory, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter:
                vals[r['Kernel_Name']].append(float(r['Counter_Value']))
    return vals


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('fetch_dir')
    ap.add_argument('write_dir')
    ap.add_argument('--size', type=int, default=4096)
    ap.add_argument('--workload', default='pipeline')
    ap.add_argument('--kernel', default=r'SplitCfg<5, 4, 128, 16, 32, 2, 8, 5(, \d+)?>, 3, 0(, \d+)?>', help='regex on the rocprofv3 kernel name')
    ap.add_argument('--commit', default=None, help='commit the profiled build was made from (recorded with the numbers)')
    ap.add_argument('--algorithmic-bytes', type=float, default=None)
    a = ap.parse_args()
    fetch = {k: v for k, v in per_dispatch(a.fetch_dir, 'FETCH_SIZE').items() if re.search(a.kernel, k)}
    write = {k: v for k, v in per_dispatch(a.write_dir, 'WRITE_SIZE').items() if re.search(a.kernel, k)}
    if not fetch or not write:
        sys.exit(f'kernel {a.kernel!r} not found in the counter files')
    name = next(iter(fetch))
    f_kib = sum(fetch[name]) / len(fetch[name])
    w_kib = sum(write[next(iter(write))]) / len(write[next(iter(write))])
    # algorithmic bytes of the launch: the 128-channel split input (hi + lo f16 = 4 B per value) with its 2 x 8-pixel halo,
    # the packed weights once, the fp32 logits out
    S = a.size
    alg = a.algorithmic_bytes or (128 * (S + 16) ** 2 * 4 + 2 * 256 * 128 * 25 * 2 * 2 + S * S * 4)
    rec = {
        'kernel': bench_name(name), 'rocprof_kernel': name.split('(')[0], 'size': S, 'workload': a.workload,
        'dispatches': len(fetch[name]),
        'profiled_at_commit': a.commit or (open(os.path.join(ROOT, 'tools', '_run', 'HEAD')).read().strip()
                                           if os.path.exists(os.path.join(ROOT, 'tools', '_run', 'HEAD')) else None),
        # bench.py reports these numbers only while csrc/conv_split.h is the file they were measured on
        'conv_split_h_sha1': __import__('hashlib').sha1(open(os.path.join(ROOT, 'topaz_amd', 'csrc', 'conv_split.h'), 'rb').read()).hexdigest(),
        'fetch_bytes_raw': f_kib * 1024, 'fetch_bytes_corrected': 2 * f_kib * 1024, 'write_bytes': w_kib * 1024,
        'traffic_bytes_per_launch': 2 * f_kib * 1024 + w_kib * 1024, 'algorithmic_bytes': alg,
        'correction': 'FETCH_SIZE x 2 (gfx950 tallies the 128-B requests of a 16 B/lane stream at 64 B; MI355X_MICROARCH.md HBM '
                      'section); WRITE_SIZE as reported',
        'source': f'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over bench.py, {os.path.basename(a.fetch_dir)} + '
                  f'{os.path.basename(a.write_dir)}',
    }
    out = os.path.join(ROOT, 'profiles', 'pmc_dominant.json')
    json.dump(rec, open(out, 'w'), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == '__main__':
    main()
