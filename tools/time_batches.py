#!/usr/bin/env python3
"""r6: AdaptiveSharpen + SSIM of 32 images at 1080p and 4K, one call per image against the batched entry points
(fnx_adaptive_sharpen_batch, fnx_ssim_batch_enqueue): python tools/time_batches.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fennec_amd  # noqa: E402
from fennec_amd import synth  # noqa: E402

ctx = fennec_amd.Context(0)
N = 32
for (W, H) in ((1920, 1080), (3840, 2160)):
    imgs = [torch.from_numpy(im).cuda() for im in synth.large_photo_batch(W, H, range(N))]
    outs = [torch.empty_like(im) for im in imgs]

    def single():
        vals = []
        for im in imgs:
            sh = ctx.AdaptiveSharpen(im, 0.5)
            ctx.ssim_enqueue(im, sh)
            vals.append(ctx.fetch_result())
        return vals

    def batched(b):
        vals = []
        for i in range(0, N, b):
            sh = ctx.sharpen_batch(imgs[i:i + b], 0.5, adaptive=True, outs=outs[i:i + b])
            ctx.ssim_batch_enqueue(imgs[i:i + b], sh)
            vals += list(ctx.fetch_results(len(sh)))
        return vals

    ref = single()
    for name, fn in (("one call per image", single), ("batches of 8", lambda: batched(8)), ("batches of 32", lambda: batched(32))):
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.4:
            v = fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            v = fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / (10 * N)
        print(f"{W}x{H} AdaptiveSharpen + SSIM, {name:20s} {dt * 1e6:7.1f} us per image  {W * H / dt / 1e9:6.2f} k MP/s  max |delta| vs single calls: {max(abs(x - y) for x, y in zip(v, ref)):.1e}", flush=True)
