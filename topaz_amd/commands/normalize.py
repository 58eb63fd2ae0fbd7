"""`topaz normalize`: 2-component Gaussian-mixture (or affine) normalisation of a set of images, the mixture fit
on the MI355X (flag surface: _spec.NORMALIZE, mirroring topaz/commands/normalize.py:16-45)."""

name = 'normalize'
help = 'scale images to zero mean / unit variance of the background component of a 2-Gaussian pixel mixture'


def add_arguments(parser=None):
    from ._spec import NORMALIZE, build_parser
    return build_parser(NORMALIZE, help, parser)


def main(args):
    from ..cuda import set_device
    from ..stats import normalize_images
    set_device(args.device)
    normalize_images(args.files, args.destdir, 0, args.scale, args.affine, args.niters, args.alpha, args.beta,
                     args.sample, args.metadata, args.format_.split(','), True, args.verbose)


if __name__ == '__main__':
    main(add_arguments().parse_args())
