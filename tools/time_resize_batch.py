#!/usr/bin/env python3
"""r6: lanczosResize of 32 4K images, one call per image against fnx_lanczos_resize_batch (8 / 16 / 32 images per launch):
python tools/time_resize_batch.py [soft]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fennec_amd  # noqa: E402
from fennec_amd import synth  # noqa: E402

soft = len(sys.argv) > 1 and sys.argv[1] == "soft"
ctx = fennec_amd.Context(0)
N, W, H = 32, 3840, 2160
if soft:
    base = ctx.GaussianBlur(ctx.GaussianBlur(torch.from_numpy(synth.noise_image(W, H, 5)).cuda(), 2.0), 1.2)
    imgs = [torch.roll(base, k * 37, 1).contiguous() for k in range(N)]
else:
    imgs = [torch.from_numpy(im).cuda() for im in synth.large_photo_batch(W, H, range(N))]
for (dw, dh, src) in ((1920, 1080, imgs), (3840, 2160, None)):
    if src is None:
        src = [ctx.lanczosResize(im, 1920, 1080) for im in imgs]
    outs = [torch.empty((dh, dw, 4), dtype=torch.uint8, device="cuda") for _ in range(N)]

    def single():
        for im in src:
            ctx.lanczosResize(im, dw, dh)

    def batched(b):
        for i in range(0, N, b):
            ctx.lanczosResizeBatch(src[i:i + b], dw, dh, outs=outs[i:i + b])

    for name, fn in (("one call per image", single), ("batches of 8", lambda: batched(8)), ("batches of 16", lambda: batched(16)),
                     ("batches of 32", lambda: batched(32))):
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.5:
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / (10 * N)
        print(f"{'soft' if soft else 'ramp'} {src[0].shape[1]}x{src[0].shape[0]} -> {dw}x{dh} {name:20s} {dt * 1e6:7.1f} us per image  [{ctx.last_kernel(fennec_amd.PROF_RESIZE)}]", flush=True)
