"""Flag tables of the hot-path commands.

The flag names, short options, types, defaults and choices are the drop-in surface of the reference CLI
(topaz/commands/{extract,denoise,denoise3d,segment,downsample,normalize}.py); they are kept as data here and the
parsers are generated from them.  Help texts are this project's own wording.  Flags that only make sense
for training are accepted (so existing command lines keep parsing) and rejected at run time.
"""
from __future__ import annotations

import argparse
from typing import Any, Dict, List, Sequence, Tuple

# (option strings, keyword arguments for add_argument)
Flag = Tuple[Sequence[str], Dict[str, Any]]

_GPUS: Flag = (('--gpus',), dict(type=int, default=0, help='start this many rank processes, one per MI355X, and shard the '
                                   'inputs over them (0: a single process, or all GPUs with -d -2)'))
_THREADS: Flag = (('-j', '--num-threads'), dict(type=int, default=0, help='host threads for torch (0: library default, <0: all cores)'))

EXTRACT: List[Flag] = [
    (('paths',), dict(nargs='*', help='micrographs to pick from; read from stdin when empty')),
    (('-m', '--model'), dict(default='resnet16', help='pretrained alias, model file, or "none" when the inputs already are score maps')),
    (('-r', '--radius'), dict(type=int, help='suppression radius in pixels')),
    (('-t', '--threshold'), dict(type=float, default=-6, help='keep picks with a log-odds score above this')),
    (('-s', '--down-scale'), dict(type=float, default=1, help='divide output coordinates by this')),
    (('-x', '--up-scale'), dict(type=float, default=1, help='multiply output coordinates by this')),
    (('--num-workers',), dict(type=int, default=0, help='ignored: suppression runs on the GPU')),
    _THREADS,
    (('-p', '--patch-size'), dict(type=int, default=0, help='score the image in tiles of this size (0: whole image)')),
    (('--batch-size',), dict(type=int, default=1, help='accepted for compatibility')),
    (('--assignment-radius',), dict(type=int, help='match radius between picks and --targets (default: the extraction radius)')),
    (('--min-radius',), dict(type=int, default=5, help='radius search: lower bound')),
    (('--max-radius',), dict(type=int, default=100, help='radius search: upper bound')),
    (('--step-radius',), dict(type=int, default=5, help='radius search: step')),
    (('--targets',), dict(help='labelled coordinates; enables the radius search / validation report')),
    (('--only-validate',), dict(action='store_true', help='print validation metrics, write no picks')),
    (('-d', '--device'), dict(type=int, default=0, help='MI355X index (no CPU mode); LOCAL_RANK under torchrun')),
    (('-o', '--output'), dict(help='pick file, or directory with --per-micrograph')),
    (('--per-micrograph',), dict(action='store_true', help='one pick file per micrograph')),
    (('--suffix',), dict(default='', help='suffix of the per-micrograph file names')),
    (('--format',), dict(choices=['coord', 'csv', 'star', 'json', 'box'], default='coord', help='pick file format')),
    (('--dims',), dict(type=int, default=2, choices=[2, 3], help='2: micrographs, 3: tomograms')),
    (('-v', '--verbose'), dict(action='store_true', help='progress on stderr')),
    _GPUS,
]

_TRAINING_ONLY = 'training option (not supported on this path)'

DENOISE: List[Flag] = [
    (('-d', '--device'), dict(type=int, default=0, help='MI355X index; LOCAL_RANK under torchrun')),
    (('micrographs',), dict(nargs='*', help='images to denoise')),
    (('-o', '--output'), dict(default='', help='output directory (default: next to the input, suffix .denoised)')),
    (('--suffix',), dict(default='', help='suffix of the output names')),
    (('--format',), dict(dest='format_', default='mrc', help='mrc, tiff, png or jpg')),
    (('--normalize',), dict(action='store_true', help='standardise the output instead of restoring mean and scale')),
    (('--stack',), dict(action='store_true', help='the input is one MRC stack')),
    (('--save-prefix',), dict(help=_TRAINING_ONLY)),
    (('--save-interval',), dict(type=int, default=10, help=_TRAINING_ONLY)),
    (('-m', '--model'), dict(nargs='+', default=['unet'], help='one or more models (outputs averaged): unet, unet-small, fcnn, affine, unet-v0.2.1, or a file')),
    (('-a', '--dir-a'), dict(help=_TRAINING_ONLY)),
    (('-b', '--dir-b'), dict(help=_TRAINING_ONLY)),
    (('--hdf',), dict(help=_TRAINING_ONLY)),
    (('--preload',), dict(action='store_true', help=_TRAINING_ONLY)),
    (('--holdout',), dict(type=float, default=0.1, help=_TRAINING_ONLY)),
    (('--lowpass',), dict(type=float, default=1, help='hard low-pass before denoising (raises: broken upstream)')),
    (('--gaussian',), dict(type=float, default=0, help='sigma of a Gaussian pre-filter')),
    (('--inv-gaussian',), dict(type=float, default=0, help='sigma of an inverse-Gaussian pre-filter')),
    (('--deconvolve',), dict(action='store_true', help='covariance deconvolution (raises: broken upstream)')),
    (('--deconv-patch',), dict(type=int, default=1, help='patches for --deconvolve')),
    (('--pixel-cutoff',), dict(type=float, default=0, help='zero pixels further than this many sigma from the mean')),
    (('-s', '--patch-size'), dict(type=int, default=1024, help='tile size (<1: whole image)')),
    (('-p', '--patch-padding'), dict(type=int, default=500, help='halo around each tile')),
    (('--method',), dict(choices=['noise2noise', 'masked'], default='noise2noise', help=_TRAINING_ONLY)),
    (('--arch',), dict(choices=['unet', 'unet-small', 'unet2', 'unet3', 'fcnet', 'fcnet2', 'affine'], default='unet', help=_TRAINING_ONLY)),
    (('--optim',), dict(choices=['adam', 'adagrad', 'sgd'], default='adagrad', help=_TRAINING_ONLY)),
    (('--lr',), dict(type=float, default=0.001, help=_TRAINING_ONLY)),
    (('--criteria',), dict(default='L2', choices=['L0', 'L1', 'L2'], help=_TRAINING_ONLY)),
    (('-c', '--crop'), dict(type=int, default=800, help=_TRAINING_ONLY)),
    (('--batch-size',), dict(type=int, default=4, help=_TRAINING_ONLY)),
    (('--num-epochs',), dict(type=int, default=100, help=_TRAINING_ONLY)),
    (('--num-workers',), dict(type=int, default=16, help=_TRAINING_ONLY)),
    _THREADS,
    _GPUS,
]

DENOISE3D: List[Flag] = [
    (('volumes',), dict(nargs='*', help='tomograms to denoise')),
    (('-o', '--output'), dict(default='', help='output directory (default: next to the input, suffix .denoised)')),
    (('--suffix',), dict(default='', help='suffix of the output names')),
    (('-m', '--model'), dict(default='unet-3d', help='unet-3d, unet-3d-10a, unet-3d-20a, or a saved model / state_dict file')),
    (('-a', '--even-train-path'), dict(help=_TRAINING_ONLY)),
    (('-b', '--odd-train-path'), dict(help=_TRAINING_ONLY)),
    (('--N-train',), dict(type=int, default=1000, help=_TRAINING_ONLY)),
    (('--N-test',), dict(type=int, default=200, help=_TRAINING_ONLY)),
    (('-c', '--crop'), dict(type=int, default=96, help=_TRAINING_ONLY)),
    (('--base-kernel-width',), dict(type=int, default=11, help='first-layer kernel width of a state_dict model')),
    (('--optim',), dict(choices=['adam', 'adagrad', 'sgd'], default='adagrad', help=_TRAINING_ONLY)),
    (('--lr',), dict(type=float, default=0.001, help=_TRAINING_ONLY)),
    (('--criteria',), dict(default='L2', choices=['L1', 'L2'], help=_TRAINING_ONLY)),
    (('--momentum',), dict(type=float, default=0.8, help=_TRAINING_ONLY)),
    (('--batch-size',), dict(type=int, default=10, help=_TRAINING_ONLY)),
    (('--num-epochs',), dict(type=int, default=500, help=_TRAINING_ONLY)),
    (('-w', '--weight_decay'), dict(type=float, default=0, help=_TRAINING_ONLY)),
    (('--save-interval',), dict(type=int, default=10, help=_TRAINING_ONLY)),
    (('--save-prefix',), dict(help=_TRAINING_ONLY)),
    (('--num-workers',), dict(type=int, default=1, help=_TRAINING_ONLY)),
    _THREADS,
    (('-g', '--gaussian'), dict(type=float, default=0, help='sigma of a Gaussian post-filter (raises: a no-op upstream)')),
    (('-s', '--patch-size'), dict(type=int, default=96, help='tile size (<1: whole volume)')),
    (('-p', '--patch-padding'), dict(type=int, default=48, help='halo around each tile')),
    (('-d', '--device'), dict(type=int, default=-2, help='-2: every visible MI355X (one rank process each; LOCAL_RANK when already under a launcher); >=0: that GPU; -1 is an error')),
    _GPUS,
]

SEGMENT: List[Flag] = [
    (('paths',), dict(nargs='+', help='images to score')),
    (('-m', '--model'), dict(default='resnet16', help='pretrained alias or model file')),
    (('-o', '--destdir'), dict(help='directory for the score maps')),
    (('-d', '--device'), dict(type=int, default=0, help='MI355X index')),
    _THREADS,
    (('-p', '--patch-size'), dict(type=int, default=None, help='score in tiles of twice this size (default: whole image)')),
    (('-v', '--verbose'), dict(action='store_true', help='progress on stdout')),
    _GPUS,
]

DOWNSAMPLE: List[Flag] = [
    (('file',), dict()),
    (('-s', '--scale'), dict(type=int, default=4, help='integer reduction factor')),
    (('-o', '--output'), dict(help='file to write')),
    (('-v', '--verbose'), dict(action='store_true', help='report the shapes')),
]


NORMALIZE: List[Flag] = [
    (('files',), dict(nargs='+')),
    (('-s', '--scale'), dict(default=1, type=int, help='shrink the images by this factor first')),
    (('--affine',), dict(action='store_true', help='plain (x - mean) / std of the whole image instead of the mixture fit')),
    (('--sample',), dict(default=10, type=int, help='fit the mixture on every n-th pixel (random subset)')),
    (('--niters',), dict(default=100, type=int, help='cap on the EM iterations of one fit')),
    (('-a', '--alpha'), dict(default=900, type=float, help='alpha of the Beta prior on the mixing proportion')),
    (('-b', '--beta'), dict(default=1, type=float, help='beta of the Beta prior on the mixing proportion')),
    (('--metadata',), dict(action='store_true', help='also write <name>.metadata.json with the fitted parameters')),
    (('-d', '--device'), dict(default=0, type=int, help='GPU to use')),
    (('-t', '--num-workers'), dict(type=int, default=0, help='accepted for compatibility (the fit runs on the GPU)')),
    (('-j', '--num-threads'), dict(type=int, default=0, help='accepted for compatibility')),
    (('-o', '--destdir'), dict(help='directory to write into')),
    (('--format',), dict(dest='format_', default='mrc', help='comma separated list of mrc, tiff, png')),
    (('-v', '--verbose'), dict(action='store_true', help='name every processed file')),
]


def build_parser(flags: List[Flag], description: str, parser: argparse.ArgumentParser = None) -> argparse.ArgumentParser:
    parser = parser or argparse.ArgumentParser(description=description)
    for names, kw in flags:
        parser.add_argument(*names, **kw)
    return parser
