"""`topaz segment` -- topaz/commands/segment.py:16-56 + segment_images (model/utils.py:71-105)."""
import os

import numpy as np
import torch

name = 'segment'
help = 'write per-pixel score maps of a trained classifier'


def add_arguments(parser=None):
    from ._spec import SEGMENT, build_parser
    return build_parser(SEGMENT, help, parser)


def segment_images(model, paths, output_dir, use_cuda, verbose, patch_size=None):
    from ..model.utils import predict_in_patches
    from ..utils.image import load_image
    os.makedirs(output_dir, exist_ok=True)
    for path in paths:
        image_name = os.path.splitext(os.path.basename(path))[0]
        image = load_image(path, make_image=False, return_header=False)
        is_3d = image.ndim == 3
        with torch.no_grad():
            X = torch.from_numpy(np.array(image)).float().unsqueeze(0).unsqueeze(0)
            if patch_size is not None:
                # (the reference passes an unsupported keyword here and crashes; this is the
                #  evident intent: patches of 2*patch_size with width//2 overlap)
                score = predict_in_patches(model, X, patch_size=patch_size * 2, is_3d=is_3d, use_cuda=True)
            else:
                score = model(X.cuda()).cpu().numpy()
            score = score[0, 0]
        out = os.path.join(output_dir, image_name)
        if verbose:
            print('# saving:', out)
        if is_3d:
            np.save(out + '.npy', score)
        else:
            from PIL import Image
            Image.fromarray(np.asarray(score, dtype=np.float32)).save(out + '.tiff', 'tiff')


def main(args):
    from ..cuda import set_device
    from ..model.factory import load_model
    from ..torch import set_num_threads
    set_num_threads(args.num_threads)
    use_cuda = set_device(args.device)
    model = load_model(args.model)
    model.eval()
    model.fill()
    model.cuda(args.device)
    if (args.patch_size is not None) and (args.patch_size <= 0):
        raise ValueError('patch size must be positive')
    segment_images(model, args.paths, args.destdir, use_cuda, args.verbose, args.patch_size)


if __name__ == '__main__':
    main(add_arguments().parse_args())
