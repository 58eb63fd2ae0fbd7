#!/usr/bin/env python
"""Does the denoise stage of micrograph i + 1 hide under the scoring stage of micrograph i?

    python tools/pipeline_probe.py [--steps 12]

The step of bench.py runs its three stages back to back on one stream; the denoise stage is ~160 launches whose deep levels
leave most CUs idle, the scoring stage five or six launches that each fill the chip for milliseconds.  Here the two stages run
from two host threads on two tpz contexts (two streams) of the SAME device, handing the denoised micrograph over a queue; the
result of every micrograph is compared bit for bit with the serial run's.
"""
from __future__ import annotations

import argparse
import os
import queue
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=12)
    ap.add_argument('--size', type=int, default=4096)
    ap.add_argument('--depth', type=int, default=2, help='denoised micrographs the denoise thread may run ahead')
    args = ap.parse_args()
    import bench
    from topaz_amd import runtime as rt
    from topaz_amd.runtime import Context, DeviceModel
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    bargs = argparse.Namespace(patch_size=1024, patch_padding=500, radius=14, threshold=-6.0)
    models = bench.build_models('pipeline')
    n = args.steps
    imgs = [torch.from_numpy(np.random.RandomState(1000 + i).randn(args.size, args.size).astype(np.float32)).to(dev)
            for i in range(min(n, 8))]
    for _ in range(2):
        bench.run_step(models, imgs[0], bargs)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    ref = [bench.run_step(models, imgs[i % len(imgs)], bargs) for i in range(n)]
    torch.cuda.synchronize(dev)
    t_serial = time.perf_counter() - t0
    print(f'serial   : {1e3 * t_serial / n:8.2f} ms / micrograph', flush=True)

    # ---- two contexts, two threads
    den, scorer = models['denoise'][0], models['score'][0]
    ctx_d = Context(0)
    s_d, s_s = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    den.model._device_model = DeviceModel(den.model._program, ctx_d)
    q = queue.Queue(maxsize=args.depth)
    out = [None] * n
    err = []

    def stage_denoise(count):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(s_d):
                ctx_d.bind_current_stream()
                for i in range(count):
                    y = den.denoise_device(imgs[i % len(imgs)], bargs.patch_size, bargs.patch_padding)
                    q.put((i, y))           # (tpz_denoise_2d returns after its stream has drained: y is complete)
        except Exception as e:              # noqa: BLE001
            err.append(e)
        q.put(None)

    def stage_score():
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(s_s):
                while True:
                    it = q.get()
                    if it is None:
                        return
                    i, y = it
                    logits = scorer(y[None, None])[0, 0]
                    out[i] = rt.nms(logits, bargs.radius, bargs.threshold)
        except Exception as e:              # noqa: BLE001
            err.append(e)

    def run(count):
        ta = threading.Thread(target=stage_denoise, args=(count,))
        tb = threading.Thread(target=stage_score)
        ta.start(); tb.start()
        ta.join(); tb.join()
        if err:
            raise err[0]

    run(3)                                   # warm the second context's pools
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    run(n)
    torch.cuda.synchronize(dev)
    t_pipe = time.perf_counter() - t0
    print(f'pipelined: {1e3 * t_pipe / n:8.2f} ms / micrograph   ({t_serial / t_pipe:.3f}x)', flush=True)
    same = all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(ref, out))
    print('pick tables identical to the serial run:', same)


if __name__ == '__main__':
    main()
