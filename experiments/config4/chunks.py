#!/usr/bin/env python3
"""Config 4 (8K AdaptiveSharpen + SSIM, FNX_SSIM_FAST) per image with results fetched three images behind -- the bench line's
flow -- against the batched entry points over chunks of 2 / 4 / 8 images, one context.  python experiments/config4/chunks.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fennec_amd  # noqa: E402
from fennec_amd import synth  # noqa: E402

ctx = fennec_amd.Context(0)
ctx.set_ssim_mode(True)
N, W, H = 8, 7680, 4320
imgs = [torch.from_numpy(im).cuda() for im in synth.large_photo_batch(W, H, range(N))]
outs = [torch.empty_like(im) for im in imgs]


def single():
    vals, pend, held = [], 0, []
    for im in imgs:
        sh = ctx.AdaptiveSharpen(im, 0.5)
        held.append(sh)                      # (ssim_enqueue's contract: the pair stays alive until its result is fetched)
        ctx.ssim_enqueue(im, sh)
        pend += 1
        if pend > 3:
            vals.append(ctx.fetch_result())
            pend -= 1
    while pend:
        vals.append(ctx.fetch_result())
        pend -= 1
    return vals


def batched(b):
    vals, pend = [], []
    for i in range(0, N, b):
        sh = ctx.sharpen_batch(imgs[i:i + b], 0.5, adaptive=True, outs=outs[i:i + b])
        ctx.ssim_batch_enqueue(imgs[i:i + b], sh)
        pend.append(len(sh))
        if len(pend) > 2:
            vals += list(ctx.fetch_results(pend.pop(0)))
    while pend:
        vals += list(ctx.fetch_results(pend.pop(0)))
    return vals


ref = single()
for rnd in range(2):
    for name, fn in (("one call per image", single), ("chunks of 2", lambda: batched(2)), ("chunks of 4", lambda: batched(4)), ("chunks of 8", lambda: batched(8))):
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 1.0:
            v = fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            v = fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / (20 * N)
        print(f"{W}x{H} {name:20s} {dt * 1e6:7.1f} us per image  {W * H / dt / 1e9:6.1f} k MP/s  max |delta| vs single calls: {max(abs(x - y) for x, y in zip(v, ref)):.1e}", flush=True)
