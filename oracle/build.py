"""Oracle (test infrastructure): compile oracle/nms_c.c with gcc into oracle/_build/."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(HERE, '_build')
LIB = os.path.join(OUT_DIR, 'libnms_oracle.so')


def build(force: bool = False) -> str:
    src = os.path.join(HERE, 'nms_c.c')
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        os.makedirs(OUT_DIR, exist_ok=True)
        subprocess.run(['gcc', '-O2', '-shared', '-fPIC', '-o', LIB, src, '-lm'], check=True)
    return LIB


if __name__ == '__main__':
    print(build(force=True))
