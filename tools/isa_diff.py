#!/usr/bin/env python
"""Per-kernel instruction statistics of a device assembly file (hipcc -S --cuda-device-only), or the kernels that differ
between two such files: the check that a refactoring of a kernel template left the generated code alone.

    python tools/isa_diff.py new.s [old.s]
"""
import re
import subprocess
import sys


def kernels(path):
    txt = open(path).read()
    out = {}
    for b in re.split(r'\n(?=_ZN3tpz\S+:)', txt):
        m = re.match(r'(_ZN3tpz\S+):', b)
        if not m:
            continue
        body = b.split('.Lfunc_end')[0]
        ins = [l for l in body.split('\n') if l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;')]
        cnt = lambda pat: sum(1 for l in ins if re.search(pat, l))
        out[m.group(1)] = dict(n=len(ins), mfma=cnt('v_mfma'), scratch=cnt('scratch_'), readlane=cnt('v_readlane'),
                               sload=cnt('s_load'), vmem=cnt(r'\t(global|flat|buffer)_'), lds=cnt(r'\tds_'))
    return out


def dem(n):
    d = subprocess.run(['c++filt', n], capture_output=True, text=True).stdout.strip()
    return re.sub(r'void tpz::|tpz::|\(.*', '', d)


if __name__ == '__main__':
    new = kernels(sys.argv[1])
    old = kernels(sys.argv[2]) if len(sys.argv) > 2 else None
    for k, v in new.items():
        if old is None or k not in old:
            print('NEW    ' if old is not None else '', dem(k)[:100], v)
        elif old[k] != v:
            print('CHANGED', dem(k)[:100], '\n    old', old[k], '\n    new', v)
    if old is not None:
        print(f'{len(new)} kernels, {sum(1 for k in new if k in old and old[k] == new[k])} unchanged')
