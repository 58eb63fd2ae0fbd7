"""Mirror of topaz/extract.py: NonMaximumSuppression / nms_iterator (:26-104), ExtractMatches /
extract_auprc / find_opt_radius (:113-204), score_images (:224-256), stream_inputs (:259-263),
extract_particles (:266-367).

Scoring and NMS run on the MI355X (one image in flight per rank); host code keeps the reference's
file handling.  Differences: no CPU path (`device` < 0 raises), `num_workers` pools are accepted and
ignored (NMS is on the GPU), and with WORLD_SIZE > 1 (torchrun) the images are sharded
`i = rank (mod world)` and the pick tables gathered to rank 0 over RCCL (topaz_amd/parallel.py).
The reference's `sys.path.join` typo (extract.py:318) is not reproduced: an existing directory passed
as `-o` receives `extracted_particles.txt`.
"""
from __future__ import annotations

import os
import sys
from typing import Iterable, Iterator, List, Tuple, Union

import numpy as np
import pandas as pd
import torch

from . import parallel
from .algorithms import non_maximum_suppression, non_maximum_suppression_3d
from .metrics import average_precision
from .model.factory import load_model
from .model.utils import get_patches, predict_in_patches
from .utils import files as file_utils
from .utils.image import load_image
from .utils.printing import report


def match_coordinates(targets: np.ndarray, preds: np.ndarray, radius: float):
    """topaz/algorithms.py:7-22 (Hungarian matching within `radius`)"""
    from scipy.optimize import linear_sum_assignment
    d2 = np.sum((preds[:, np.newaxis] - targets[np.newaxis]) ** 2, 2)
    cost = d2 - radius * radius
    cost[cost > 0] = 0
    pred_index, target_index = linear_sum_assignment(cost)
    cost = cost[pred_index, target_index]
    dist = np.zeros(len(preds))
    dist[pred_index] = np.sqrt(d2[pred_index, target_index])
    pred_index = pred_index[cost < 0]
    assignment = np.zeros(len(preds), dtype=np.float32)
    assignment[pred_index] = 1
    return assignment, dist


class NonMaximumSuppression:
    def __init__(self, radius: int, threshold: float, dims: int = 2, patch_size=64, patch_overlap=32, verbose=False):
        self.radius, self.threshold, self.dims = radius, threshold, dims
        self.patch_size, self.patch_overlap, self.verbose = patch_size, patch_overlap, verbose

    def __call__(self, args):
        nms = non_maximum_suppression if self.dims == 2 else non_maximum_suppression_3d
        name, score = args
        if self.verbose:
            report(f'Scoring {name}')
        if not self.patch_size:
            s, c = nms(score, self.radius, threshold=self.threshold)
            return name, s, c
        # patched variant (extract.py:42-72); the CLI never enables it (extract.py:333)
        t = torch.as_tensor(np.asarray(score))
        y, x = t.shape[-2:]
        z = t.shape[-3] if self.dims == 3 else None
        patches = get_patches(t, self.patch_size, self.patch_overlap, is_3d=(self.dims == 3))
        step = self.patch_size - self.patch_overlap * 2
        scores_list, coords_list, idx = [], [], 0
        for i in range(0, y, step):
            for j in range(0, x, step):
                for k in (range(0, z, step) if self.dims == 3 else [None]):
                    ps, pc = nms(patches[idx].numpy(), self.radius, threshold=self.threshold)
                    ps, pc = crop_translate_coords_scores(ps, pc, self.patch_size, self.patch_overlap, j, i, k)
                    scores_list.append(ps)
                    coords_list.append(pc)
                    idx += 1
        s = np.concatenate(scores_list, axis=0) if scores_list else np.array([])
        c = np.concatenate(coords_list, axis=0) if coords_list else np.array([])
        return name, s, c


def crop_translate_coords_scores(scores, coords, patch_size, patch_overlap, x, y, z=None):
    within = np.all(np.logical_and(patch_overlap <= coords, coords < patch_size + patch_overlap), axis=-1)
    coords, scores = coords[within], scores[within]
    coords[:, -1] += x
    coords[:, -2] += y
    if z is not None:
        coords[:, -3] += z
    return scores, coords


def nms_iterator(paths_scores, radius, threshold, pool=None, dims=2, patch_size=0, patch_overlap=0, verbose=False):
    process = NonMaximumSuppression(radius, threshold, dims=dims, patch_size=patch_size, patch_overlap=patch_overlap,
                                    verbose=verbose)
    for name, score in paths_scores:
        yield process((name, score))


def extract_auprc(targets, scores, radius, threshold, match_radius=None, pool=None, dims=2):
    N, mse, hits, preds = 0, 0, [], []
    for image_name, score in scores.items():
        target = targets.loc[targets.image_name == image_name][['x_coord', 'y_coord']].values
        if dims == 2:
            s, coords = non_maximum_suppression(score, radius, threshold=threshold)
        else:
            s, coords = non_maximum_suppression_3d(score, radius * 2, threshold=threshold)
        assignment, dist = match_coordinates(target, coords, radius if match_radius is None else match_radius)
        mse += np.sum(dist[assignment == 1] ** 2)
        hits.append(assignment)
        preds.append(s)
        N += len(target)
    hits, preds = np.concatenate(hits, 0), np.concatenate(preds, 0)
    return average_precision(hits, preds, N=N), np.sqrt(mse / hits.sum()), int(hits.sum()), N


def find_opt_radius(targets, target_scores, threshold, lo=0, hi=200, step=10, match_radius=None, pool=None, dims=2):
    auprc = np.zeros(hi + 1) - 1
    for r in range(lo, hi + 1, step):
        au, rmse, recall, n = extract_auprc(targets, target_scores, r, threshold, match_radius=match_radius, dims=2)
        auprc[r] = au
        print('# radius={}, auprc={}, rmse={}, recall={}, targets={}'.format(r, au, rmse, recall, n))
    r = int(np.argmax(auprc))
    return r, auprc[r]


def score_images(model, paths: Iterable[str], device: int = 0, patch_size: int = 0, batch_size: int = 1,
                 keep_on_device: bool = False) -> Iterator[Tuple[str, np.ndarray]]:
    """generator of (path, scores).  keep_on_device=True yields device tensors (no PCIe round trip
    before the NMS)."""
    if model is not None and model != 'none':
        if device is not None and device < 0:
            raise RuntimeError('topaz_amd has no CPU path: use -d >= 0 (an MI355X)')
        torch.cuda.set_device(device)
        model = load_model(model)
        model.eval()
        model.fill()
        model.cuda(device)
        for path in paths:
            image = load_image(path, make_image=False, return_header=False)
            is_3d = image.ndim == 3
            x = torch.from_numpy(np.array(image)).float().unsqueeze(0).unsqueeze(0)     # image.copy() upstream
            if patch_size:
                overlap = model.width // 2
                scores = predict_in_patches(model, x, patch_size + 2 * overlap, is_3d=is_3d, use_cuda=True)[0, 0]
            else:
                with torch.no_grad():
                    scores = model(x.cuda(device))[0, 0]
                if not keep_on_device:
                    scores = scores.cpu().numpy()
            yield path, scores
    else:
        for path in paths:
            yield path, load_image(path, make_image=False, return_header=False)


def stream_inputs(f):
    for line in f:
        line = line.strip()
        if len(line) > 0:
            yield line


def extract_particles(paths: List[str], model, device: int, batch_size: int, threshold: float, radius: int,
                      num_workers: int, targets: str, min_radius: int, max_radius: int, step: int, match_radius: int,
                      patch_size, only_validate: bool, output: str, per_micrograph: bool, suffix: str, out_format: str,
                      up_scale: float, down_scale: float, dims=2, verbose: bool = False):
    report('Beginning extraction')
    rank, local_rank, world = parallel.init_from_env()
    if world > 1:
        device = local_rank
    paths = list(stream_inputs(sys.stdin)) if len(paths) == 0 else list(paths)
    my_idx = parallel.shard_indices(len(paths), rank, world)
    stream = score_images(model, [paths[i] for i in my_idx], device=device, patch_size=patch_size,
                          batch_size=batch_size, keep_on_device=(targets is None))
    radius = radius if radius is not None else -1

    if targets is not None:
        if world > 1:
            raise NotImplementedError('--targets radius search is single-process')
        scores = {k: v for k, v in stream}
        stream = scores.items()
        targets = pd.read_csv(targets, sep='\t')
        target_scores = {name: scores[name] for name in targets.image_name.unique() if name in scores}
        if radius < 0:
            report('Finding optimal radius for extraction')
            radius, auprc = find_opt_radius(targets, target_scores, threshold, lo=min_radius, hi=max_radius, step=step,
                                            match_radius=match_radius, dims=dims)
            report(f'Optimal radius found: {radius} with AUPRC: {auprc}')
        else:
            au, rmse, recall, n = extract_auprc(targets, target_scores, radius, threshold, match_radius=match_radius,
                                                dims=dims)
            print('# radius={}, auprc={}, rmse={}, recall={}, targets={}'.format(radius, au, rmse, recall, n))
    elif radius < 0:
        raise Exception('Must specify targets for choosing the extraction radius if extraction radius is not provided')

    if not only_validate:
        scale = up_scale / down_scale
        f = None
        if not per_micrograph:
            if output is not None and os.path.isdir(output):
                output = os.path.join(output, 'extracted_particles.txt')
        elif not os.path.isdir(output):
            os.makedirs(os.path.dirname(output) or '.', exist_ok=True)
            output_dir = os.path.join(os.path.dirname(output), 'COORDS')
            os.makedirs(output_dir, exist_ok=True)       # the reference forgets this and fails
        else:
            output_dir = output

        gathered = []      # (global image index, path, scores, coords) kept for the single-file mode
        for local_i, (path, score, coords) in enumerate(nms_iterator(stream, radius, threshold, dims=dims,
                                                                      verbose=verbose)):
            basename = os.path.basename(path)
            name, ext = os.path.splitext(basename)
            if verbose:
                report(f'Extracted {len(score)} particles from {name}')
            coords = np.round(coords * scale).astype(int) if scale != 1 else coords
            if per_micrograph:
                out_path = os.path.join(output_dir, name + suffix + '.' + out_format)
                cols = {'image_name': name, 'x_coord': coords[:, 0], 'y_coord': coords[:, 1]}
                if dims == 3:
                    cols['z_coord'] = coords[:, 2]
                cols['score'] = score
                with open(out_path, 'w') as fo:
                    file_utils.write_table(fo, pd.DataFrame(cols), format=out_format, image_ext=ext)
            else:
                gathered.append((my_idx[local_i], name, score, coords))

        if not per_micrograph:
            tables = {i: (n, s, c) for i, n, s, c in gathered}
            if world > 1:
                dev = torch.device('cuda', local_rank)
                got = parallel.gather_pick_tables([g[0] for g in gathered],
                                                  [torch.from_numpy(np.asarray(g[2], dtype=np.float32)) for g in gathered],
                                                  [torch.from_numpy(np.asarray(g[3], dtype=np.int32)) for g in gathered], dev)
                if rank == 0:
                    tables = {i: (os.path.splitext(os.path.basename(paths[i]))[0], s.numpy(), c.numpy())
                              for i, (s, c) in got.items()}
            if rank == 0:
                f = sys.stdout if output is None else open(output, 'w')
                z_string = '\tz_coord' if dims == 3 else ''
                print(f'image_name\tx_coord\ty_coord{z_string}\tscore', file=f)
                for i in sorted(tables):
                    name, score, coords = tables[i]
                    for k in range(len(score)):
                        z_coord = f'\t{coords[k, 2]}' if dims == 3 else ''
                        print(f'{name}\t{coords[k, 0]}\t{coords[k, 1]}{z_coord}\t{score[k]}', file=f)
                if f is not sys.stdout:
                    f.close()
    report('Extraction complete')
