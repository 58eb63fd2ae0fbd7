"""`topaz downsample`: truncated-DFT reduction of one image on the MI355X (flag surface: _spec.DOWNSAMPLE,
mirroring topaz/commands/downsample.py:10-26)."""

name = 'downsample'
help = 'shrink an image by truncating its Fourier spectrum'


def add_arguments(parser=None):
    from ._spec import DOWNSAMPLE, build_parser
    return build_parser(DOWNSAMPLE, help, parser)


def main(args):
    from ..utils.image import downsample_file
    downsample_file(args.file, args.scale, args.output, args.verbose)


if __name__ == '__main__':
    main(add_arguments().parse_args())
