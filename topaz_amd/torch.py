"""Host thread-count helper with the surface of topaz/torch.py:5-12."""
import os

import torch


def set_num_threads(num_threads: int) -> int:
    """0 keeps torch's default, a negative value means every core; returns the count that is in effect"""
    n = (os.cpu_count() or 1) if num_threads < 0 else num_threads
    if n:
        torch.set_num_threads(n)
    return n
