#!/usr/bin/env python3
"""1080p -> 4K upscale, batches of 4 / 8 / 16: resize_fused_kernel (the default route for upscales) against the matrix-pipe
route (form resize_mfma = 2), on the downscaled ramp and on soft content.  python experiments/config3/upscale_route.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fennec_amd  # noqa: E402
from fennec_amd import synth  # noqa: E402

ctx = fennec_amd.Context(0)
N, W, H = 32, 3840, 2160
for content in ("ramp", "soft"):
    if content == "soft":
        base = ctx.GaussianBlur(ctx.GaussianBlur(torch.from_numpy(synth.noise_image(W, H, 5)).cuda(), 2.0), 1.2)
        imgs = [torch.roll(base, k * 37, 1).contiguous() for k in range(N)]
    else:
        imgs = [torch.from_numpy(im).cuda() for im in synth.large_photo_batch(W, H, range(N))]
    src = [ctx.lanczosResize(im, 1920, 1080) for im in imgs]
    outs = [torch.empty((H, W, 4), dtype=torch.uint8, device="cuda") for _ in range(N)]
    ref = None
    for route in ("default", "2"):
        ctx = fennec_amd.Context(0)                       # (the plans are cached per ctx and built under the form in force)
        ctx.set_form("resize_mfma", None if route == "default" else route)
        for b in (1, 4, 8, 16):
            def fn():
                for i in range(0, N, b):
                    ctx.lanczosResizeBatch(src[i:i + b], W, H, outs=outs[i:i + b])
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.4:
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / (10 * N)
            print(f"{content} 1080p -> 4K route {route:8s} batches of {b:2d}: {dt * 1e6:6.1f} us per image [{ctx.last_kernel(fennec_amd.PROF_RESIZE)}]", flush=True)
        if ref is None:
            ref = [o.clone() for o in outs[:4]]
        else:
            print("   same bytes as the default route:", all(torch.equal(a, b_) for a, b_ in zip(ref, outs[:4])))
