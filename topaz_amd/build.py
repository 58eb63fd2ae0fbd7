"""Build libtopaz_hip.so (gfx950) in-tree with hipcc.

    python -m topaz_amd.build [--force] [--jobs N]

One object per .hip translation unit (compiled in parallel, rebuilt only when the content of the
source, of a header it includes or the flags changed), linked into topaz_amd/libtopaz_hip.so.  hipcc cross-compiles for gfx950
without a GPU, so this runs on the CPU-only build container as well as on the MI355X box.
"""
from __future__ import annotations

import argparse
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
BUILD = os.path.join(CSRC, 'build')
LIB = os.path.join(HERE, 'libtopaz_hip.so')
ARCH = 'gfx950'


def _hipcc() -> str:
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found (set HIPCC or install ROCm)')


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))


FLAGS = [f'--offload-arch={ARCH}', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function']


def _all_headers():
    hdrs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith('.h')]
    hdrs.append(os.path.join(HERE, '..', 'include', 'topaz_hip.h'))
    return [h for h in hdrs if os.path.exists(h)]


def _fingerprint(src: str, obj: str) -> str:
    """content hash of everything the object is compiled from: the source, the in-tree headers its depfile lists
    (hipcc -MD; every header when there is no depfile yet) and the flags.  File times say nothing after the tree has
    been copied to another box -- a stale-looking tree was recompiled there on every build()."""
    dep = obj[:-2] + '.d'
    hdrs = _all_headers()
    if os.path.exists(dep):
        names = open(dep).read().replace('\\\n', ' ').split()[1:]
        listed = {os.path.basename(n) for n in names if n.endswith('.h')}
        hdrs = [h for h in hdrs if os.path.basename(h) in listed]
    h = hashlib.sha1(' '.join(FLAGS).encode())
    for f in [src] + hdrs:
        h.update(os.path.basename(f).encode())
        h.update(open(f, 'rb').read())
    return h.hexdigest()


def _compile(hipcc: str, src: str, obj: str) -> None:
    cmd = [hipcc] + FLAGS + ['-I', CSRC, '-MD', '-MF', obj[:-2] + '.d', '-c', src, '-o', obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'hipcc failed on {src}:\n{r.stdout}\n{r.stderr}')
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    with open(obj[:-2] + '.sha1', 'w') as f:          # (after the depfile exists: the fingerprint covers what was included)
        f.write(_fingerprint(src, obj))


def build(force: bool = False, jobs: int | None = None, verbose: bool = True) -> str:
    hipcc = _hipcc()
    os.makedirs(BUILD, exist_ok=True)
    todo, objs = [], []
    for f in _sources():
        src = os.path.join(CSRC, f)
        obj = os.path.join(BUILD, f[:-4] + '.o')
        objs.append(obj)
        mark = obj[:-2] + '.sha1'
        fresh = os.path.exists(obj) and os.path.exists(mark) and open(mark).read().strip() == _fingerprint(src, obj)
        if force or not fresh:
            todo.append((src, obj))
    if todo:
        if verbose:
            print(f'[topaz_amd.build] compiling {len(todo)} translation unit(s) for {ARCH}', file=sys.stderr)
        jobs = jobs or min(len(todo), os.cpu_count() or 4)
        with ThreadPoolExecutor(max_workers=jobs) as ex:
            list(ex.map(lambda so: _compile(hipcc, *so), todo))
    link_mark = os.path.join(BUILD, 'link.sha1')
    link_fp = hashlib.sha1(''.join(open(o[:-2] + '.sha1').read() for o in objs).encode()).hexdigest()
    if todo or not os.path.exists(LIB) or not os.path.exists(link_mark) or open(link_mark).read().strip() != link_fp:
        cmd = [hipcc, f'--offload-arch={ARCH}', '-shared', '-fPIC', '-o', LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'link failed:\n{r.stdout}\n{r.stderr}')
        with open(link_mark, 'w') as f:
            f.write(link_fp)
        if verbose:
            print(f'[topaz_amd.build] linked {LIB}', file=sys.stderr)
    return LIB


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--force', action='store_true')
    ap.add_argument('--jobs', type=int, default=None)
    a = ap.parse_args()
    print(build(force=a.force, jobs=a.jobs))
