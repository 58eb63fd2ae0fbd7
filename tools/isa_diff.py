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


def loop_stats(path, pat):
    """the K loop of every kernel whose demangled name matches `pat`: the innermost span [label, backward branch to it] that
    contains every v_mfma of the function"""
    txt = open(path).read()
    for b in re.split(r'\n(?=_ZN3tpz\S+:)', txt):
        m = re.match(r'(_ZN3tpz\S+):', b)
        if not m or not re.search(pat, dem(m.group(1))):
            continue
        lines = [l for l in b.split('.Lfunc_end')[0].split('\n') if l.strip() and not l.startswith('\t.') and not l.startswith('\t;')]
        labels = {l.split(':')[0]: i for i, l in enumerate(lines) if re.match(r'\.LBB\d+_\d+:', l)}
        idx = [i for i, l in enumerate(lines) if 'v_mfma' in l]
        best = None
        for i, l in enumerate(lines):
            mb = re.search(r'\ts_c?branch\S*\s+(\.LBB\d+_\d+)', l)
            if mb and i > idx[-1] and labels.get(mb.group(1), 1 << 30) < idx[0]:
                span = (labels[mb.group(1)], i)
                if best is None or span[1] - span[0] < best[1] - best[0]:
                    best = span
        if best is None:
            print(dem(m.group(1))[:90], 'no loop found')
            continue
        body = [l for l in lines[best[0]:best[1] + 1] if l.startswith('\t')]
        cnt = lambda p: sum(1 for l in body if re.search(p, l))
        print(dem(m.group(1))[:90], {'body': len(body), 'mfma': cnt('v_mfma'), 'readlane': cnt('v_readlane'), 'writelane': cnt('v_writelane'),
                                     'salu': cnt(r'\ts_'), 'sload': cnt('s_load'), 'valu': cnt(r'\tv_') - cnt('v_mfma'), 'ds': cnt(r'\tds_'),
                                     'vmem': cnt(r'\tbuffer_|\tglobal_'), 'branch': cnt('s_cbranch|s_branch'), 'waitcnt': cnt('s_waitcnt')})
