#!/usr/bin/env python3
"""GPU-stage-only rate of CompressBatch: decoded 4K sources RESIDENT on the device, the C++ pool
(fennec_CompressBatchNRGBA) runs compressJPEGOptimal per item on the device and hands back the files.
python tools/time_batch_native.py [n_images] [workers...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fennec_amd  # noqa: E402,F401
from fennec_amd import batch, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ws = [int(x) for x in sys.argv[2:]] or [1, 2, 4, 8]
W, H = 3840, 2160
imgs = [torch.from_numpy(im).cuda() for im in synth.large_photo_batch(W, H, range(n))]
torch.cuda.synchronize()
for workers in ws:
    batch.compress_batch_native(imgs[:workers], workers=workers)            # contexts' scratch, clocks
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        res, files, summ = batch.compress_batch_native(imgs, workers=workers)
    dt = (time.perf_counter() - t0) / reps
    print(f"{n} x 4K, {workers} workers: {n / dt:7.1f} images/s  ({dt / n * 1e3:.2f} ms per image; quality {res[0].Quality}, "
          f"{sum(len(f) for f in files) / n / 1e6:.2f} MB per file, AvgSSIM {summ.AvgSSIM:.5f})", flush=True)
