"""`topaz preprocess`: downsample and normalise in one step -- the same flags and worker as `topaz normalize`
(topaz/commands/preprocess.py:4-14)."""
from .normalize import add_arguments, main  # noqa: F401

name = 'preprocess'
help = 'shrink and normalise images in one step'
