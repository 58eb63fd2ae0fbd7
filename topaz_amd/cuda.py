"""topaz/cuda.py:16-32 set_device -- without the CPU fallback: the MI355X path has no CPU mode."""
import torch


def set_device(device, error=False, warn=True):
    if device < 0:
        raise RuntimeError('topaz_amd has no CPU path: choose an MI355X with -d >= 0')
    torch.cuda.set_device(device)
    return True
