"""`topaz` entry point for the MI355X hot path: the sub-command registry of topaz/main.py:53-148 restricted to
the commands of the path (segment, extract, downsample, normalize, preprocess, denoise, denoise3d); `@file` argument expansion kept (main.py:55)."""
import argparse
import sys


def main(argv=None):
    from . import _version
    from .commands import denoise, denoise3d, downsample, extract, normalize, preprocess, segment
    parser = argparse.ArgumentParser(prog='topaz', formatter_class=argparse.RawDescriptionHelpFormatter,
                                     fromfile_prefix_chars='@',
                                     description='topaz on MI355X: particle extraction and denoising (inference hot path)')
    parser.add_argument('--version', action='version', version=_version.__version__)
    subparsers = parser.add_subparsers(title='commands', metavar='<command>')
    subparsers.required = True
    for module in (segment, extract, downsample, normalize, preprocess, denoise, denoise3d):
        p = subparsers.add_parser(module.name, help=module.help, fromfile_prefix_chars='@')
        module.add_arguments(p)
        p.set_defaults(func=module.main)
    args = parser.parse_args(argv)
    return args.func(args)


if __name__ == '__main__':
    sys.exit(main())
