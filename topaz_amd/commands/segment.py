"""`topaz segment` -- topaz/commands/segment.py:16-56 + segment_images (model/utils.py:71-105)."""
import os

import numpy as np
import torch

name = 'segment'
help = 'write per-pixel score maps of a trained classifier'


def add_arguments(parser=None):
    from ._spec import SEGMENT, build_parser
    return build_parser(SEGMENT, help, parser)


class MapSink:
    """where score maps go: <destdir>/<micrograph name>.tiff (float32, 2-D) or .npy (volumes) -- model/utils.py:96-105"""

    def __init__(self, destdir: str, verbose: bool):
        os.makedirs(destdir, exist_ok=True)
        self.destdir, self.verbose = destdir, verbose

    def write(self, source_path: str, scores: np.ndarray) -> str:
        stem = os.path.join(self.destdir, os.path.splitext(os.path.basename(source_path))[0])
        if self.verbose:
            print('# saving:', stem)
        if scores.ndim == 3:
            np.save(stem + '.npy', scores)
            return stem + '.npy'
        from PIL import Image
        Image.fromarray(np.ascontiguousarray(scores, dtype=np.float32)).save(stem + '.tiff', 'tiff')
        return stem + '.tiff'


def score_map(model, pixels: np.ndarray, patch_size=None) -> np.ndarray:
    """per-pixel (per-voxel) logits of one micrograph / tomogram: the filled model over the whole array or, with `patch_size`,
    over tiles of twice that size that overlap by the receptive field (the evident intent of model/utils.py:88-90, which passes
    predict_in_patches a keyword it does not take and crashes upstream -- SURVEY 3.1)"""
    from ..model.utils import predict_in_patches
    x = torch.as_tensor(np.ascontiguousarray(pixels), dtype=torch.float32)[None, None]
    if patch_size is not None:
        return predict_in_patches(model, x, patch_size=2 * patch_size, is_3d=(pixels.ndim == 3), use_cuda=True)[0, 0]
    with torch.no_grad():
        return model(x.cuda())[0, 0].cpu().numpy()


def segment_images(model, paths, output_dir, use_cuda, verbose, patch_size=None):
    """topaz.model.utils.segment_images (model/utils.py:71-105): one score map file per input"""
    from ..utils.image import load_image
    sink = MapSink(output_dir, verbose)
    for path in paths:
        sink.write(path, score_map(model, load_image(path, make_image=False, return_header=False), patch_size))


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
