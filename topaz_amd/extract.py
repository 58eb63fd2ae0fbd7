"""`topaz extract` on the MI355X: score micrographs, suppress non-maxima, write pick tables.

Drop-in surface of topaz/extract.py -- the names callers import keep their arguments and return values
(NonMaximumSuppression / nms_iterator :26-104, extract_auprc / find_opt_radius :135-204, score_images :224-256,
stream_inputs :259-263, extract_particles :266-367) -- over a different engine:

  ImageFeed      a reader thread decodes micrograph i+1 straight into a pinned staging slot (tpz_stage) and queues its
                 H2D copy while the GPU scores micrograph i; the host never waits for a copy.
  Scorer         the filled network as one HIP model (load once, one forward per image); score maps stay in HBM.
  suppression    tpz_nms_2d / _3d on the device map; only the pick table crosses PCIe.
  RadiusSearch   `--targets`: score maps cached on the device once, one device NMS + one Hungarian matching per radius.
  PickSink       the three output shapes (one TSV, per-micrograph tables, stdout).

Differences from the reference, all deliberate: there is no CPU path (`device` < 0 raises); worker pools are accepted and
ignored (the suppression is on the GPU); with WORLD_SIZE > 1 the micrographs are dealt round-robin to the ranks and the
pick tables gathered to rank 0 over RCCL (topaz_amd/parallel.py); an existing directory given as `-o` receives
`extracted_particles.txt` (upstream: `sys.path.join` typo, extract.py:318) and `COORDS/` is created (upstream forgets).
The patched suppression branch of NonMaximumSuppression cannot run upstream (extract.py:55 unpacks three values from a
two-tuple; the default 64/32 tiling has step 0): it is implemented here as written-to-be, tile by tile on the device.
"""
from __future__ import annotations

import os
import queue
import sys
import threading
from typing import Dict, Iterable, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import pandas as pd
import torch

from . import mrc
from . import parallel
from . import runtime as rt
from .algorithms import match_coordinates, non_maximum_suppression, non_maximum_suppression_3d
from .metrics import average_precision
from .model.factory import load_model
from .model.utils import predict_in_patches
from .utils import files as file_utils
from .utils.image import load_image
from .utils.printing import report

__all__ = ['NonMaximumSuppression', 'crop_translate_coords_scores', 'nms_iterator', 'extract_auprc', 'find_opt_radius',
           'score_images', 'stream_inputs', 'extract_particles', 'match_coordinates']


# ---------------------------------------------------------------------------------------------------------------------
# suppression
# ---------------------------------------------------------------------------------------------------------------------
def _suppress(score, radius: int, threshold: float, dims: int) -> Tuple[np.ndarray, np.ndarray]:
    if dims == 3:
        return non_maximum_suppression_3d(score, radius, threshold=threshold)
    return non_maximum_suppression(score, radius, threshold=threshold)


def crop_translate_coords_scores(scores, coords, patch_size, patch_overlap, x, y, z=None):
    """keep the picks of a tile that lie in its core [overlap, overlap + patch_size) on every axis and move them to
    image coordinates (tile origin x, y[, z]); coords columns are (x, y[, z])"""
    coords = np.asarray(coords)
    core = ((coords >= patch_overlap) & (coords < patch_size + patch_overlap)).all(axis=-1)
    moved = coords[core].copy()
    for column, origin in enumerate((x, y) if z is None else (x, y, z)):
        moved[:, column] += origin
    return np.asarray(scores)[core], moved


class NonMaximumSuppression:
    """callable (name, score map) -> (name, scores, coords); whole image when `patch_size` is falsy (what the CLI uses)"""

    def __init__(self, radius: int, threshold: float, dims: int = 2, patch_size=64, patch_overlap=32, verbose: bool = False):
        self.radius, self.threshold, self.dims = radius, threshold, dims
        self.patch_size, self.patch_overlap, self.verbose = patch_size, patch_overlap, verbose

    def __call__(self, item):
        name, score = item
        if self.verbose:
            report(f'Scoring {name}')
        if not self.patch_size:
            s, c = _suppress(score, self.radius, self.threshold, self.dims)
            return name, s, c
        step = self.patch_size - 2 * self.patch_overlap
        if step <= 0:
            raise ValueError(f'patch_size {self.patch_size} leaves no core after removing 2 x {self.patch_overlap} overlap')
        t = (score if torch.is_tensor(score) else torch.as_tensor(np.asarray(score))).float()
        ov, size = self.patch_overlap, self.patch_size
        # halo of -inf: the padding must neither be picked nor suppress picks next to the border
        padded = torch.nn.functional.pad(t[None], (ov, ov) * self.dims, value=float('-inf'))[0]
        shape = tuple(t.shape)
        kept_s, kept_c = [], []
        for i in range(0, shape[-2], step):
            for j in range(0, shape[-1], step):
                for k in (range(0, shape[-3], step) if self.dims == 3 else (None,)):
                    tile = padded[i:i + size, j:j + size] if k is None else padded[k:k + size, i:i + size, j:j + size]
                    s, c = _suppress(tile.contiguous(), self.radius, self.threshold, self.dims)
                    s, c = crop_translate_coords_scores(s, c, step, ov, j - ov, i - ov, None if k is None else k - ov)
                    kept_s.append(s)
                    kept_c.append(c)
        if not kept_s:
            return name, np.array([]), np.array([])
        return name, np.concatenate(kept_s, axis=0), np.concatenate(kept_c, axis=0)


def nms_iterator(paths_scores, radius, threshold, pool=None, dims=2, patch_size=0, patch_overlap=0, verbose=False):
    """(name, score map) pairs -> (name, scores, coords); `pool` is accepted and ignored (the GPU is the pool)"""
    suppress = NonMaximumSuppression(radius, threshold, dims=dims, patch_size=patch_size, patch_overlap=patch_overlap,
                                     verbose=verbose)
    return (suppress(item) for item in paths_scores)


# ---------------------------------------------------------------------------------------------------------------------
# `--targets`: validation and radius search
# ---------------------------------------------------------------------------------------------------------------------
class RadiusSearch:
    """Score maps (kept wherever they are -- device tensors stay in HBM) against a table of labelled coordinates.
    evaluate(r) = one suppression per micrograph at radius r (doubled for tomograms, like ExtractMatches :125),
    Hungarian matching of picks to targets within `match_radius` (default r), pooled average precision."""

    def __init__(self, targets: pd.DataFrame, scores: Dict[str, object], threshold: float, match_radius=None, dims: int = 2):
        self.threshold, self.match_radius, self.dims = threshold, match_radius, dims
        self.pairs = [(score, targets.loc[targets.image_name == name, ['x_coord', 'y_coord']].values)
                      for name, score in scores.items()]

    def evaluate(self, radius) -> Tuple[float, float, int, int]:
        sq_err, n_targets, hit_flags, hit_scores = 0.0, 0, [], []
        for score, target in self.pairs:
            s, coords = _suppress(score, radius * 2 if self.dims == 3 else radius, self.threshold, self.dims)
            matched, dist = match_coordinates(target, coords, radius if self.match_radius is None else self.match_radius)
            sq_err += float(np.sum(dist[matched == 1] ** 2))
            hit_flags.append(matched)
            hit_scores.append(s)
            n_targets += len(target)
        hits, preds = np.concatenate(hit_flags), np.concatenate(hit_scores)
        n_hit = hits.sum()
        return average_precision(hits, preds, N=n_targets), np.sqrt(sq_err / n_hit), int(n_hit), n_targets

    @staticmethod
    def line(radius, result) -> str:
        au, rmse, recall, n = result
        return '# radius={}, auprc={}, rmse={}, recall={}, targets={}'.format(radius, au, rmse, recall, n)

    def best(self, lo: int, hi: int, step: int) -> Tuple[int, float]:
        curve = np.full(hi + 1, -1.0)
        for r in range(lo, hi + 1, step):
            res = self.evaluate(r)
            curve[r] = res[0]
            print(self.line(r, res))
        r = int(np.argmax(curve))
        return r, curve[r]


def extract_auprc(targets, scores, radius, threshold, match_radius=None, pool=None, dims=2):
    return RadiusSearch(targets, scores, threshold, match_radius, dims).evaluate(radius)


def find_opt_radius(targets, target_scores, threshold, lo=0, hi=200, step=10, match_radius=None, pool=None, dims=2):
    return RadiusSearch(targets, target_scores, threshold, match_radius, dims).best(lo, hi, step)


# ---------------------------------------------------------------------------------------------------------------------
# scoring
# ---------------------------------------------------------------------------------------------------------------------
class ImageFeed:
    """Iterate (path, device tensor [H, W] or [D, H, W]) with the NEXT micrograph's disk read, fp32 conversion and H2D
    copy running under the consumer's work on the current one.  A reader thread decodes into the pinned buffer of a
    free staging slot and queues the copy; the consumer acquires the slot on its stream, uses the tensor, and the slot is
    released (for re-use two images later) when the consumer asks for the next item."""

    DEPTH = 3

    def __init__(self, paths: Sequence[str], ctx: 'rt.Context', headers: bool = False):
        """headers=True: items are (path, tensor, MRC header or None, extended header or None)"""
        self.paths, self.ctx, self.headers = list(paths), ctx, headers
        self.stage: Optional[rt.Stage] = None
        self.slot_bytes = 0
        self.ready: 'queue.Queue' = queue.Queue(maxsize=self.DEPTH - 1)
        self.free: 'queue.Queue' = queue.Queue()
        self.thread = threading.Thread(target=self._reader, daemon=True)
        self._retired: List['rt.Stage'] = []      # outgrown rings: closed by the consumer when the feed ends
        self._stop = threading.Event()           # the consumer left (done, or an exception): the reader must not block

    def _ensure_stage(self, nbytes: int) -> None:
        # (called from the reader thread before any slot of a new, larger ring is handed out)
        if self.stage is None or nbytes > self.slot_bytes:
            old = self.stage
            self.slot_bytes = (nbytes + (1 << 20) - 1) & ~((1 << 20) - 1)
            self.stage = rt.Stage(self.ctx, self.slot_bytes, self.DEPTH)
            while not self.free.empty():
                self.free.get_nowait()
            for k in range(self.DEPTH):
                self.free.put((self.stage, k))
            if old is not None:
                self._retired.append(old)  # closed by the consumer when the feed ends (never from this thread: tpz_stage_free
                                           # synchronises the ctx stream the consumer is enqueuing on)

    def _get_free(self):
        """a free slot, or None once the consumer has left"""
        while not self._stop.is_set():
            try:
                return self.free.get(timeout=0.1)
            except queue.Empty:
                pass
        return None

    def _put_ready(self, item) -> bool:
        while not self._stop.is_set():
            try:
                self.ready.put(item, timeout=0.1)
                return True
            except queue.Full:
                pass
        return False

    def _reader(self) -> None:
        try:
            torch.cuda.set_device(self.ctx.device)
            class _Left(Exception):
                pass

            held = []

            def pinned(shape):
                """the pinned buffer of a free staging slot for an image of that shape (the ring grows when it must)"""
                nbytes = int(np.prod(shape)) * 4
                if self.stage is None or nbytes > self.slot_bytes:
                    # wait until every slot of the old ring came back, then grow
                    if self.stage is not None:
                        for _ in range(self.DEPTH):
                            if self._get_free() is None:
                                raise _Left()
                    self._ensure_stage(nbytes)
                slot = self._get_free()
                if slot is None:
                    raise _Left()
                # the slot's previous upload may still be queued (a consumer that only enqueues runs ahead of the copy
                # stream): the pinned buffer is refilled only after that copy has read it
                slot[0].wait(slot[1])
                held[:] = [slot, nbytes]
                return slot[0].host_array(slot[1], shape)

            for path in self.paths:
                header = extended = None
                try:
                    # an MRC file is decoded straight into pinned memory (float32: read INTO it); anything else is loaded and
                    # converted into it
                    got = mrc.read_into(path, pinned) if os.path.splitext(path)[1] == '.mrc' else None
                    if got is not None:
                        host, header, extended = got
                    else:
                        loaded = load_image(path, make_image=False)
                        image, header, extended = loaded if isinstance(loaded, tuple) else (loaded, None, None)
                        image = np.asarray(image)
                        host = pinned(image.shape)
                        np.copyto(host, image, casting='unsafe')
                except _Left:
                    return
                (stage, k), nbytes = held
                stage.upload(k, nbytes)
                if not self._put_ready((path, stage, k, host.shape, header, extended)):
                    return
            self._put_ready(None)
        except BaseException as e:                                            # surface reader failures in the consumer
            self._put_ready(e)

    def __iter__(self) -> Iterator[Tuple[str, torch.Tensor]]:
        self.thread.start()
        held = None
        try:
            while True:
                if held is not None:            # the consumer came back for more: it is done with the previous tensor
                    held[0].release(held[1])
                    self.free.put(held)
                    held = None
                item = self.ready.get()
                if item is None:
                    return
                if isinstance(item, BaseException):
                    raise item
                path, stage, k, shape, header, extended = item
                stage.acquire(k)
                held = (stage, k)
                if self.headers:
                    yield path, stage.device_tensor(k, shape), header, extended
                else:
                    yield path, stage.device_tensor(k, shape)
        finally:
            if held is not None:
                held[0].release(held[1])
            # the consumer is leaving (exhausted, early exit or exception): unblock and join the reader, then close the rings
            # from THIS thread (the one that enqueues on the ctx stream)
            self._stop.set()
            if self.thread.is_alive():
                self.thread.join(timeout=30)
            if not self.thread.is_alive():
                for st in self._retired:
                    st.close()
                self._retired = []
                if self.stage is not None:
                    self.stage.close()
                    self.stage = None


class Scorer:
    """the filled scoring network on one MI355X"""

    def __init__(self, model, device: int = 0):
        if device is not None and device < 0:
            raise RuntimeError('topaz_amd has no CPU path: use -d >= 0 (an MI355X)')
        torch.cuda.set_device(device)
        self.device = device
        self.model = load_model(model)
        self.model.eval()
        self.model.fill()
        self.model.cuda(device)
        self.ctx = self.model.device_model.ctx

    def __call__(self, image: torch.Tensor, patch_size: int = 0):
        """[H, W] / [D, H, W] device tensor -> logits of the same shape (device tensor; float64 numpy when patched, like
        predict_in_patches upstream)"""
        x = image[None, None]
        if patch_size:
            halo = self.model.width // 2
            return predict_in_patches(self.model, x, patch_size + 2 * halo, is_3d=(image.dim() == 3), use_cuda=True)[0, 0]
        with torch.no_grad():
            return self.model(x)[0, 0]


def score_images(model, paths: Iterable[str], device: int = 0, patch_size: int = 0, batch_size: int = 1,
                 keep_on_device: bool = False) -> Iterator[Tuple[str, np.ndarray]]:
    """generator of (path, score map).  `model` None / 'none': the inputs already are score maps and pass through.
    keep_on_device=True yields device tensors (no PCIe round trip before the suppression)."""
    paths = list(paths)
    if model is None or model == 'none':
        for path in paths:
            yield path, load_image(path, make_image=False, return_header=False)
        return
    scorer = Scorer(model, device)
    for path, image in ImageFeed(paths, scorer.ctx):
        scores = scorer(image, patch_size)
        if torch.is_tensor(scores) and not keep_on_device:
            scores = scores.cpu().numpy()
        yield path, scores


def stream_inputs(f) -> Iterator[str]:
    return (line.strip() for line in f if line.strip())


# ---------------------------------------------------------------------------------------------------------------------
# output
# ---------------------------------------------------------------------------------------------------------------------
class PickSink:
    """where pick tables go: one TSV (file or stdout, rank 0 writes it after the gather) or one table per micrograph
    (each rank writes its own files)"""

    def __init__(self, output: Optional[str], per_micrograph: bool, suffix: str, out_format: str, dims: int):
        self.per_micrograph, self.suffix, self.format, self.dims = per_micrograph, suffix, out_format, dims
        self.rows: List[Tuple[int, str, np.ndarray, np.ndarray]] = []
        if per_micrograph:
            if os.path.isdir(output):
                self.dir = output
            else:
                parent = os.path.dirname(output)
                self.dir = os.path.join(parent, 'COORDS')
                os.makedirs(self.dir, exist_ok=True)
            self.path = None
        else:
            self.dir = None
            self.path = os.path.join(output, 'extracted_particles.txt') if (output is not None and os.path.isdir(output)) \
                else output

    def add(self, index: int, path: str, scores: np.ndarray, coords: np.ndarray) -> None:
        name, ext = os.path.splitext(os.path.basename(path))
        if not self.per_micrograph:
            self.rows.append((index, name, scores, coords))
            return
        cols = {'image_name': name, 'x_coord': coords[:, 0], 'y_coord': coords[:, 1]}
        if self.dims == 3:
            cols['z_coord'] = coords[:, 2]
        cols['score'] = scores
        with open(os.path.join(self.dir, name + self.suffix + '.' + self.format), 'w') as f:
            file_utils.write_table(f, pd.DataFrame(cols), format=self.format, image_ext=ext)

    def finish(self, paths: Sequence[str], rank: int, world: int, local_rank: int) -> None:
        if self.per_micrograph:
            return
        tables = {i: (name, s, c) for i, name, s, c in self.rows}
        if world > 1:
            dev = parallel.collective_device(local_rank)
            got = parallel.gather_pick_tables([r[0] for r in self.rows],
                                              [torch.from_numpy(np.asarray(r[2], dtype=np.float32)) for r in self.rows],
                                              [torch.from_numpy(np.asarray(r[3], dtype=np.int32)) for r in self.rows], dev)
            if rank == 0:
                tables = {i: (os.path.splitext(os.path.basename(paths[i]))[0], s.numpy(), c.numpy()) for i, (s, c) in got.items()}
        if rank != 0:
            return
        # rows formatted by the library's host-side writer (tpz_format_picks: the f-string's text -- a float32 score with the
        # digits of its float64 value -- byte for byte; tests/test_cpu_host.py), not by a Python loop over the picks
        out = sys.stdout if self.path is None else open(self.path, 'w')
        try:
            print('image_name\tx_coord\ty_coord' + ('\tz_coord' if self.dims == 3 else '') + '\tscore', file=out)
            for i in sorted(tables):
                name, scores, coords = tables[i]
                out.write(file_utils.format_pick_rows(name, coords, scores, self.dims).decode())
        finally:
            if out is not sys.stdout:
                out.close()


# ---------------------------------------------------------------------------------------------------------------------
# the command
# ---------------------------------------------------------------------------------------------------------------------
def extract_particles(paths: List[str], model, device: int, batch_size: int, threshold: float, radius: int,
                      num_workers: int, targets: str, min_radius: int, max_radius: int, step: int, match_radius: int,
                      patch_size, only_validate: bool, output: str, per_micrograph: bool, suffix: str, out_format: str,
                      up_scale: float, down_scale: float, dims=2, verbose: bool = False):
    report('Beginning extraction')
    rank, local_rank, world = parallel.init_from_env()
    if world > 1:
        device = parallel.rank_device(local_rank)
    if len(paths):
        paths = list(paths)
    elif parallel.under_launcher() and os.environ.get('TOPAZ_AMD_INPUT_LIST'):
        # a rank of `topaz extract --gpus N < list`: the launcher read stdin once for all ranks (main.py).  Only a rank process
        # honours the variable, and only once: a stale / exported copy must not silently replace stdin elsewhere
        with open(os.environ.pop('TOPAZ_AMD_INPUT_LIST')) as f:
            paths = list(stream_inputs(f))
    else:
        paths = list(stream_inputs(sys.stdin))
    mine = parallel.shard_indices(len(paths), rank, world)
    maps = score_images(model, [paths[i] for i in mine], device=device, patch_size=patch_size, batch_size=batch_size,
                        keep_on_device=True)
    radius = -1 if radius is None else radius

    if targets is not None:
        # every score map is needed at once (they stay in HBM); the table is keyed by the paths as given (extract.py:284-290)
        if world > 1:
            raise NotImplementedError('--targets (radius search / validation) runs in a single process')
        maps = dict(maps)
        table = pd.read_csv(targets, sep='\t')
        search = RadiusSearch(table, {n: maps[n] for n in table.image_name.unique() if n in maps}, threshold, match_radius, dims)
        if radius < 0:
            report('Finding optimal radius for extraction')
            radius, auprc = search.best(min_radius, max_radius, step)
            report(f'Optimal radius found: {radius} with AUPRC: {auprc}')
        else:
            print(search.line(radius, search.evaluate(radius)))
        maps = maps.items()
    elif radius < 0:
        raise Exception('Must specify targets for choosing the extraction radius if extraction radius is not provided')

    if not only_validate:
        sink = PickSink(output, per_micrograph, suffix, out_format, dims)
        scale = up_scale / down_scale
        for k, (path, scores, coords) in enumerate(nms_iterator(maps, radius, threshold, dims=dims, verbose=verbose)):
            if verbose:
                report(f'Extracted {len(scores)} particles from {os.path.splitext(os.path.basename(path))[0]}')
            if scale != 1:
                coords = np.round(coords * scale).astype(int)
            sink.add(mine[k], path, scores, coords)
        sink.finish(paths, rank, world, local_rank)
    report('Extraction complete')
