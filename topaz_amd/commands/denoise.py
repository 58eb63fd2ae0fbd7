"""`topaz denoise` -- inference flags of topaz/commands/denoise.py:19-69 (training flags are accepted and
rejected with a clear message: training is out of scope).  Unlike the reference (which loads `args.arch`
and ignores `-m`, commands/denoise.py:106) the models named by `-m` are the ones used."""
import argparse
import sys

from .. import denoise as dn
from ..cuda import set_device
from ..denoise import Denoise, denoise_stack, denoise_stream

name = 'denoise'
help = 'denoise micrographs with various denoising algorithms'


def add_arguments(parser=None):
    if parser is None:
        parser = argparse.ArgumentParser(help)
    parser.add_argument('-d', '--device', default=0, type=int, help='which MI355X to use (default: 0); under torchrun each rank uses LOCAL_RANK')
    parser.add_argument('micrographs', nargs='*', help='micrographs to denoise')
    parser.add_argument('-o', '--output', default='', help='directory to save denoised micrographs')
    parser.add_argument('--suffix', default='', help='add this suffix to each output file name. if no output directory is specified, denoised micrographs are written to the same location as the input with a default suffix of ".denoised" (default: none)')
    parser.add_argument('--format', dest='format_', default='mrc', help='output format for the images (default: mrc)')
    parser.add_argument('--normalize', action='store_true', help='normalize the micrographs')
    parser.add_argument('--stack', action='store_true', help='denoise a MRC stack rather than list of micorgraphs')
    parser.add_argument('--save-prefix', help='(training) not supported')
    parser.add_argument('--save-interval', default=10, type=int, help='(training) not supported')
    parser.add_argument('-m', '--model', nargs='+', default=['unet'], help='use pretrained denoising model(s). can accept arguments for multiple models the outputs of which will be averaged. pretrained model options are: unet, unet-small, fcnn, affine. to use older unet version specify unet-v0.2.1 (default: unet)')
    parser.add_argument('-a', '--dir-a', help='(training) not supported')
    parser.add_argument('-b', '--dir-b', help='(training) not supported')
    parser.add_argument('--hdf', help='(training) not supported')
    parser.add_argument('--preload', action='store_true', help='(training) not supported')
    parser.add_argument('--holdout', type=float, default=0.1, help='(training) not supported')
    parser.add_argument('--lowpass', type=float, default=1, help='lowpass filter micrographs by this amount (in pixels) before applying the denoising filter (default: no lowpass filtering)')
    parser.add_argument('--gaussian', type=float, default=0, help='Gaussian filter micrographs with this standard deviation (in pixels) before applying the denoising filter (default: 0)')
    parser.add_argument('--inv-gaussian', type=float, default=0, help='Inverse Gaussian filter micrographs with this standard deviation (in pixels) before applying the denoising filter (default: 0)')
    parser.add_argument('--deconvolve', action='store_true', help='apply optimal Gaussian deconvolution filter to each micrograph before denoising')
    parser.add_argument('--deconv-patch', type=int, default=1, help='apply spatial covariance correction to micrograph to this many patches (default: 1)')
    parser.add_argument('--pixel-cutoff', type=float, default=0, help='set pixels >= this number of standard deviations away from the mean to the mean. only used when set > 0 (default: 0)')
    parser.add_argument('-s', '--patch-size', type=int, default=1024, help='denoises micrographs in patches of this size. not used if < 1 (default: 1024)')
    parser.add_argument('-p', '--patch-padding', type=int, default=500, help='padding around each patch to remove edge artifacts (default: 500)')
    parser.add_argument('--method', choices=['noise2noise', 'masked'], default='noise2noise', help='(training) not supported')
    parser.add_argument('--arch', choices=['unet', 'unet-small', 'unet2', 'unet3', 'fcnet', 'fcnet2', 'affine'], default='unet', help='(training) not supported')
    parser.add_argument('--optim', choices=['adam', 'adagrad', 'sgd'], default='adagrad', help='(training) not supported')
    parser.add_argument('--lr', default=0.001, type=float, help='(training) not supported')
    parser.add_argument('--criteria', default='L2', choices=['L0', 'L1', 'L2'], help='(training) not supported')
    parser.add_argument('-c', '--crop', type=int, default=800, help='(training) not supported')
    parser.add_argument('--batch-size', type=int, default=4, help='(training) not supported')
    parser.add_argument('--num-epochs', default=100, type=int, help='(training) not supported')
    parser.add_argument('--num-workers', default=16, type=int, help='(training) not supported')
    parser.add_argument('-j', '--num-threads', type=int, default=0, help='number of threads for pytorch, 0 uses pytorch defaults, <0 uses all cores (default: 0)')
    return parser


def main(args):
    from ..torch import set_num_threads
    from .. import parallel
    set_num_threads(args.num_threads)
    _, local_rank, world = parallel.init_from_env()
    device = local_rank if world > 1 else args.device
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
                          args.patch_padding, normalize, use_cuda)


if __name__ == '__main__':
    main(add_arguments().parse_args())
