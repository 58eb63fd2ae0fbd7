"""`topaz denoise3d` -- inference flags of topaz/commands/denoise3d.py:14-58.  `-d -2` (all GPUs, DataParallel
upstream) means here: run under torchrun, one rank per GPU, volumes sharded over ranks."""
import argparse
import sys

from ..denoise import Denoise3D, denoise_tomogram_stream

name = 'denoise3d'
help = 'denoise 3D volumes with various denoising algorithms'


def add_arguments(parser=None):
    if parser is None:
        parser = argparse.ArgumentParser(help)
    parser.add_argument('volumes', nargs='*', help='volumes to denoise')
    parser.add_argument('-o', '--output', default='', help='directory to save denoised volumes')
    parser.add_argument('--suffix', default='', help='add this suffix to each output file name. if no output directory is specified, denoised tomograms are written to the same location as the input with a default suffix of ".denoised" (default: none)')
    parser.add_argument('-m', '--model', default='unet-3d', help='use pretrained denoising model. accepts path to a previously saved model or one of the provided pretrained models. pretrained model options are: unet-3d, unet-3d-10a, unet-3d-20a (default: unet-3d)')
    parser.add_argument('-a', '--even-train-path', help='(training) not supported')
    parser.add_argument('-b', '--odd-train-path', help='(training) not supported')
    parser.add_argument('--N-train', type=int, default=1000, help='(training) not supported')
    parser.add_argument('--N-test', type=int, default=200, help='(training) not supported')
    parser.add_argument('-c', '--crop', type=int, default=96, help='(training) not supported')
    parser.add_argument('--base-kernel-width', type=int, default=11, help='width of the base convolutional filter kernel in the U-net model (default: 11)')
    parser.add_argument('--optim', choices=['adam', 'adagrad', 'sgd'], default='adagrad', help='(training) not supported')
    parser.add_argument('--lr', default=0.001, type=float, help='(training) not supported')
    parser.add_argument('--criteria', default='L2', choices=['L1', 'L2'], help='(training) not supported')
    parser.add_argument('--momentum', type=float, default=0.8, help='(training) not supported')
    parser.add_argument('--batch-size', type=int, default=10, help='(training) not supported')
    parser.add_argument('--num-epochs', type=int, default=500, help='(training) not supported')
    parser.add_argument('-w', '--weight_decay', type=float, default=0, help='(training) not supported')
    parser.add_argument('--save-interval', default=10, type=int, help='(training) not supported')
    parser.add_argument('--save-prefix', help='(training) not supported')
    parser.add_argument('--num-workers', type=int, default=1, help='(training) not supported')
    parser.add_argument('-j', '--num-threads', type=int, default=0, help='number of threads for pytorch, 0 uses pytorch defaults, <0 uses all cores (default: 0)')
    parser.add_argument('-g', '--gaussian', type=float, default=0, help='standard deviation of Gaussian filter postprocessing, 0 means no postprocessing (default: 0)')
    parser.add_argument('-s', '--patch-size', type=int, default=96, help='denoises volumes in patches of this size. not used if <1 (default: 96)')
    parser.add_argument('-p', '--patch-padding', type=int, default=48, help='padding around each patch to remove edge artifacts (default: 48)')
    parser.add_argument('-d', '--device', type=int, default=-2, help='compute device to use; -2 (default): LOCAL_RANK under torchrun, else device 0')
    return parser


def main(args):
    import torch
    from ..torch import set_num_threads
    from .. import parallel
    set_num_threads(args.num_threads)
    _, local_rank, world = parallel.init_from_env()
    if args.device == -1:
        print('ERROR: Invalid device id or format', file=sys.stderr)
        sys.exit(1)
    device = local_rank if (world > 1 or args.device == -2) else args.device
    if device >= torch.cuda.device_count():
        print('ERROR: Invalid device id or format', file=sys.stderr)
        sys.exit(1)
    torch.cuda.set_device(device)
    print(f'# using device={device} with cuda=True', file=sys.stderr)
    if args.even_train_path is not None or args.odd_train_path is not None:
        raise NotImplementedError('training denoising models is out of scope of the MI355X hot path')
    print('# Warning: no denoising model will be used' if args.model == 'none' else '# Loading model:' + str(args.model),
          file=sys.stderr)
    denoiser = Denoise3D(args.model, True, dims=3)
    total = len(args.volumes)
    if total < 1:
        return
    print(f'# denoising {total} tomograms with patch size={args.patch_size} and padding={args.patch_padding}',
          file=sys.stderr)
    return denoise_tomogram_stream(volumes=args.volumes, model=denoiser, output_path=args.output, suffix=args.suffix,
                                   gaus=args.gaussian, patch_size=args.patch_size, padding=args.patch_padding,
                                   verbose=True, use_cuda=True)


if __name__ == '__main__':
    main(add_arguments().parse_args())
