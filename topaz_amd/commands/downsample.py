"""`topaz downsample` -- topaz/commands/downsample.py:10-26 (truncated-DFT downsample, on the MI355X)."""
import argparse

name = 'downsample'
help = 'downsample micrographs with truncated DFT'


def add_arguments(parser=None):
    if parser is None:
        parser = argparse.ArgumentParser()
    parser.add_argument('file')
    parser.add_argument('-s', '--scale', default=4, type=int, help='downsampling factor (default: 4)')
    parser.add_argument('-o', '--output', help='output file')
    parser.add_argument('-v', '--verbose', action='store_true', help='print info')
    return parser


def main(args):
    from ..utils.image import downsample_file
    downsample_file(args.file, args.scale, args.output, args.verbose)


if __name__ == '__main__':
    main(add_arguments().parse_args())
