import sys


def report(*args):
    """topaz/utils/printing.py:5-6"""
    print('#', *args, file=sys.stderr)
