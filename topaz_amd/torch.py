"""topaz/torch.py:5-12"""
import torch


def set_num_threads(num_threads):
    if num_threads < 0:
        from multiprocessing import cpu_count
        num_threads = cpu_count()
    if num_threads > 0:
        torch.set_num_threads(num_threads)
    return num_threads
