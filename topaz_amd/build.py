"""Build libtopaz_hip.so (gfx950) in-tree with hipcc.

    python -m topaz_amd.build [--force] [--jobs N]

One object per .hip translation unit (compiled in parallel, rebuilt only when the source or a
header is newer), linked into topaz_amd/libtopaz_hip.so.  hipcc cross-compiles for gfx950
without a GPU, so this runs on the CPU-only build container as well as on the MI355X box.
"""
from __future__ import annotations

import argparse
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


def _headers_mtime() -> float:
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    hdrs.append(os.path.join(HERE, '..', 'include', 'topaz_hip.h'))
    return max(os.path.getmtime(h) for h in hdrs if os.path.exists(h))


def _deps_mtime(obj: str, fallback: float) -> float:
    """newest in-tree header the object was compiled from (hipcc -MD depfile); every header when unknown"""
    dep = obj[:-2] + '.d'
    if not os.path.exists(dep):
        return fallback
    names = open(dep).read().replace('\\\n', ' ').split()[1:]
    root = os.path.abspath(os.path.join(HERE, '..'))
    times = [os.path.getmtime(n) for n in names
             if n.endswith('.h') and os.path.abspath(n).startswith(root) and os.path.exists(n)]
    return max(times) if times else fallback


def _compile(hipcc: str, src: str, obj: str) -> None:
    cmd = [hipcc, f'--offload-arch={ARCH}', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function',
           '-I', CSRC, '-MD', '-MF', obj[:-2] + '.d', '-c', src, '-o', obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'hipcc failed on {src}:\n{r.stdout}\n{r.stderr}')
    if r.stderr.strip():
        sys.stderr.write(r.stderr)


def build(force: bool = False, jobs: int | None = None, verbose: bool = True) -> str:
    hipcc = _hipcc()
    os.makedirs(BUILD, exist_ok=True)
    hdr_t = _headers_mtime()
    todo, objs = [], []
    for f in _sources():
        src = os.path.join(CSRC, f)
        obj = os.path.join(BUILD, f[:-4] + '.o')
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), _deps_mtime(obj, hdr_t)):
            todo.append((src, obj))
    if todo:
        if verbose:
            print(f'[topaz_amd.build] compiling {len(todo)} translation unit(s) for {ARCH}', file=sys.stderr)
        jobs = jobs or min(len(todo), os.cpu_count() or 4)
        with ThreadPoolExecutor(max_workers=jobs) as ex:
            list(ex.map(lambda so: _compile(hipcc, *so), todo))
    if todo or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [hipcc, f'--offload-arch={ARCH}', '-shared', '-fPIC', '-o', LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'link failed:\n{r.stdout}\n{r.stderr}')
        if verbose:
            print(f'[topaz_amd.build] linked {LIB}', file=sys.stderr)
    return LIB


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--force', action='store_true')
    ap.add_argument('--jobs', type=int, default=None)
    a = ap.parse_args()
    print(build(force=a.force, jobs=a.jobs))
