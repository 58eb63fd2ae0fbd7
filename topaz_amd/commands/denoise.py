"""`topaz denoise` -- inference flags of topaz/commands/denoise.py:19-69 (training flags are accepted and
rejected with a clear message: training is out of scope).  Unlike the reference (which loads `args.arch`
and ignores `-m`, commands/denoise.py:106) the models named by `-m` are the ones used."""
import sys

from .. import denoise as dn
from ..cuda import set_device
from ..denoise import Denoise, denoise_stack, denoise_stream

name = 'denoise'
help = 'denoise micrographs with pretrained or user-supplied networks'


def add_arguments(parser=None):
    from ._spec import DENOISE, build_parser
    return build_parser(DENOISE, help, parser)


def main(args):
    from ..torch import set_num_threads
    from .. import parallel
    set_num_threads(args.num_threads)
    _, local_rank, world = parallel.init_from_env()
    device = parallel.rank_device(local_rank) if world > 1 else args.device
    use_cuda = set_device(device)
    print(f'# using device={device} with cuda={use_cuda}', file=sys.stderr)
    if (args.dir_a is not None and args.dir_b is not None) or (args.hdf is not None):
        raise NotImplementedError('training denoising models is out of scope of the MI355X hot path')
    models = []
    for arg in args.model:
        print('# Warning: no denoising model will be used' if arg == 'none' else '# Loading model:' + str(arg),
              file=sys.stderr)
        if arg != 'none':
            models.append(Denoise(arg, use_cuda))
    normalize = True if args.format_ in ['png', 'jpg'] else args.normalize
    gaus = dn.GaussianDenoise(args.gaussian, use_cuda=use_cuda) if args.gaussian > 0 else None
    inv_gaus = dn.InvGaussianFilter(args.inv_gaussian, use_cuda=use_cuda) if args.inv_gaussian > 0 else None
    if len(args.micrographs) < 1:
        return
    if args.stack:
        return denoise_stack(args.micrographs[0], args.output, models, args.lowpass, args.pixel_cutoff, gaus, inv_gaus,
                             args.deconvolve, args.deconv_patch, args.patch_size, args.patch_padding, normalize, use_cuda)
    return denoise_stream(args.micrographs, args.output, args.format_, args.suffix, models, args.lowpass,
                          args.pixel_cutoff, gaus, inv_gaus, args.deconvolve, args.deconv_patch, args.patch_size,
                          args.patch_padding, normalize, use_cuda, return_images=False)


if __name__ == '__main__':
    main(add_arguments().parse_args())
