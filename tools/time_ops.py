#!/usr/bin/env python3
"""Per-op timing at 4K on device-resident images: us per call (wall clock over repeated calls after a
clock pre-warm, stream-synchronised), algorithmic GB/s (SURVEY 8(d) byte counts) and the fraction
of the 8 TB/s HBM roofline.  python tools/time_ops.py [W H]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fennec_amd  # noqa: E402
from fennec_amd import synth  # noqa: E402

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
ctx = fennec_amd.Context(0)
S = 4.0 * W * H
imgs = [torch.from_numpy(synth.large_photo(W, H, k)).cuda() for k in range(8)]
blur = [ctx.GaussianBlur(i, 2.0) for i in imgs]
half = [ctx.lanczosResize(i, W // 2, H // 2) for i in imgs[:2]]
pal = np.random.default_rng(1).integers(0, 256, size=(256, 4), dtype=np.uint8)
pal[:, 3] = 255
y, cb, cr = (torch.from_numpy(p).cuda() for p in synth.rgb_to_ycbcr_planes(imgs[0].cpu().numpy(), 2))
prep = ctx.ssim_fast_prepare(imgs[0])
ctx.sync()
torch.cuda.synchronize()

OPS = [
    ("GaussianBlur sigma=2 (fast)", lambda k: ctx.GaussianBlur(imgs[k], 2.0), 2 * S, False),
    ("GaussianBlur sigma=2 (exact)", lambda k: ctx.GaussianBlur(imgs[k], 2.0, exact=True), 2 * S, False),
    ("GaussianBlur sigma=3 (R=9, fast)", lambda k: ctx.GaussianBlur(imgs[k], 3.0), 2 * S, False),
    ("GaussianBlur sigma=4 (R=12, fast)", lambda k: ctx.GaussianBlur(imgs[k], 4.0), 2 * S, False),
    ("GaussianBlur sigma=4 (R=12, exact)", lambda k: ctx.GaussianBlur(imgs[k], 4.0, exact=True), 2 * S, False),
    ("GaussianBlur sigma=5 (R=15, fast)", lambda k: ctx.GaussianBlur(imgs[k], 5.0), 2 * S, False),
    ("GaussianBlur sigma=5 (R=15, exact)", lambda k: ctx.GaussianBlur(imgs[k], 5.0, exact=True), 2 * S, False),
    ("GaussianBlur sigma=5.3 (R=16, fast)", lambda k: ctx.GaussianBlur(imgs[k], 5.3), 2 * S, False),
    ("GaussianBlur sigma=5.3 (R=16, exact)", lambda k: ctx.GaussianBlur(imgs[k], 5.3, exact=True), 2 * S, False),
    ("GaussianBlur sigma=6 (R=18, fast)", lambda k: ctx.GaussianBlur(imgs[k], 6.0), 2 * S, False),
    ("GaussianBlur sigma=10 (R=30, fast)", lambda k: ctx.GaussianBlur(imgs[k], 10.0), 2 * S, False),
    ("GaussianBlur sigma=10 (R=30, exact)", lambda k: ctx.GaussianBlur(imgs[k], 10.0, exact=True), 2 * S, False),
    ("GaussianBlur sigma=20 (R=60, fast)", lambda k: ctx.GaussianBlur(imgs[k], 20.0), 2 * S, False),
    ("GaussianBlur sigma=21 (R=63, generic fp64)", lambda k: ctx.GaussianBlur(imgs[k], 21.0), 2 * S, False),
    ("gaussianBlur3x3", lambda k: ctx.blur3x3(imgs[k]), 2 * S, False),
    ("Sharpen 0.5", lambda k: ctx.Sharpen(imgs[k], 0.5), 2 * S, False),
    ("AdaptiveSharpen 0.5", lambda k: ctx.AdaptiveSharpen(imgs[k], 0.5), 2 * S, False),
    ("ApplyOrientation 2 (flipH)", lambda k: ctx.ApplyOrientation(imgs[k], 2), 2 * S, False),
    ("ApplyOrientation 6 (rot90)", lambda k: ctx.ApplyOrientation(imgs[k], 6), 2 * S, False),
    ("boxDownsample -> 512x288", lambda k: ctx.boxDownsample(imgs[k], 512, 288), S, False),
    ("lanczosResize -> 1/2", lambda k: ctx.lanczosResize(imgs[k], W // 2, H // 2), 1.25 * S, False),
    ("lanczosResize 1/2 -> full", lambda k: ctx.lanczosResize(half[k % 2], W, H), 1.25 * S, False),
    ("SSIMFast", lambda k: ctx.SSIMFast(imgs[k], blur[k]), 2 * S, True),
    ("SSIMFast against prepared", lambda k: prep.against(blur[k]), S, True),
    ("SSIM (full resolution)", lambda k: ctx.SSIM(imgs[k], blur[k]), 2 * S, True),
    ("MSSSIM", lambda k: ctx.MSSSIM(imgs[k], blur[k]), 3.33 * S, True),
    ("Analyze", lambda k: ctx.Analyze(imgs[k]), S, True),
    ("isOpaque", lambda k: ctx.isOpaque(imgs[k]), S, True),
    ("applyPalette 256 + quantized", lambda k: ctx.applyPalette(imgs[k], pal), 2.25 * S, False),
    ("ycbcrToNRGBA 4:2:0", lambda k: ctx.ycbcrToNRGBA(y, cb, cr, 2), 1.375 * S, False),
]
print(f"{W}x{H}, device-resident, one call at a time (launch + sync latency included)")
print(f"{'op':38s} {'us/call':>9s} {'GB/s':>8s} {'of 8 TB/s':>9s}")
for name, fn, nbytes, syncs in OPS:
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.25:
        fn(0)
        ctx.sync()
    n = 40
    t0 = time.perf_counter()
    for i in range(n):
        fn(i % 8)
    ctx.sync()
    dt = (time.perf_counter() - t0) / n
    print(f"{name:38s} {dt * 1e6:9.1f} {nbytes / dt / 1e9:8.0f} {nbytes / dt / 8e12:9.3f}")
