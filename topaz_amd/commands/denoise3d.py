"""`topaz denoise3d` -- inference flags of topaz/commands/denoise3d.py:14-58.  `-d -2` (all GPUs, DataParallel
upstream) means here: `topaz` starts one rank process per visible GPU (main._ranks_to_launch; or run it under
torchrun) and the volumes are sharded over the ranks."""
import sys

from ..denoise import Denoise3D, denoise_tomogram_stream

name = 'denoise3d'
help = 'denoise tomograms tile by tile with a 3-D U-Net'


def add_arguments(parser=None):
    from ._spec import DENOISE3D, build_parser
    return build_parser(DENOISE3D, help, parser)


def main(args):
    import torch
    from ..torch import set_num_threads
    from .. import parallel
    set_num_threads(args.num_threads)
    _, local_rank, world = parallel.init_from_env()
    if args.device == -1:
        print('ERROR: Invalid device id or format', file=sys.stderr)
        sys.exit(1)
    device = parallel.rank_device(local_rank) if (world > 1 or args.device == -2) else args.device
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
