#!/usr/bin/env python
"""Register / spill / LDS statistics of every kernel in a HIP translation unit, from the compiler's own metadata
(hipcc -S): the check that a scheduling change did not push a kernel into scratch.

    python tools/isa_stats.py topaz_amd/csrc/conv_split_inst_a.hip [...]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def stats(src):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, 'k.s')
        subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-I', os.path.join(ROOT, 'topaz_amd', 'csrc'),
                        '-S', '--cuda-device-only', '-o', out, src], check=True, stderr=subprocess.DEVNULL)
        txt = open(out).read()
    rows = []
    for blk in txt.split('  - .agpr_count:')[1:]:
        get = lambda k: int(re.search(r'\.' + k + r':\s+(\d+)', blk).group(1))
        name = re.search(r'\.name:\s+(\S+)', blk).group(1)
        dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r'void tpz::|tpz::|\(.*', '', dem)
        rows.append((dem, get('vgpr_count'), get('vgpr_spill_count'), get('sgpr_spill_count'),
                     get('private_segment_fixed_size'), txt.count('v_mfma')))
    return rows


if __name__ == '__main__':
    for src in sys.argv[1:]:
        for dem, v, vs, ss, scr, _ in stats(src):
            print(f'{v:4d} vgpr  {vs:3d} vspill  {ss:3d} sspill  {scr:5d} B scratch   {dem[:110]}')
