"""`topaz extract`: score micrographs, suppress non-maxima, write pick tables (flag surface: _spec.EXTRACT,
mirroring topaz/commands/extract.py:12-67)."""

from ..extract import extract_particles

name = 'extract'
help = 'score micrographs with a trained classifier and pick particles by non-maximum suppression'


def add_arguments(parser=None):
    from ._spec import EXTRACT, build_parser
    return build_parser(EXTRACT, help, parser)


def main(args):
    from ..torch import set_num_threads
    set_num_threads(args.num_threads)
    extract_particles(args.paths, args.model, args.device, args.batch_size, args.threshold, args.radius, args.num_workers,
                      args.targets, args.min_radius, args.max_radius, args.step_radius, args.assignment_radius,
                      args.patch_size, args.only_validate, args.output, args.per_micrograph, args.suffix, args.format,
                      args.up_scale, args.down_scale, dims=args.dims, verbose=args.verbose)


if __name__ == '__main__':
    main(add_arguments().parse_args())
