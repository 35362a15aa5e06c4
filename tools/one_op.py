#!/usr/bin/env python3
"""Run ONE op repeatedly on device-resident images (for rocprofv3 passes).
python tools/one_op.py adaptive|sharpen|blur3|ssim|ssimfast|resize_down|resize_up|msssim [W H] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fennec_amd  # noqa: E402
from fennec_amd import synth  # noqa: E402

op = sys.argv[1]
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (7680, 4320)
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 12
ctx = fennec_amd.Context(0)
imgs = [torch.from_numpy(synth.large_photo(W, H, k)).cuda() for k in range(3)]
if os.environ.get("ONE_OP_SOFT") == "1":      # blurred noise: photograph-like statistics (no exact ties)
    imgs = [ctx.GaussianBlur(ctx.GaussianBlur(torch.from_numpy(synth.noise_image(W, H, 5 + k)).cuda(), 2.0), 1.2) for k in range(3)]
other = [ctx.AdaptiveSharpen(i, 0.5) for i in imgs]
half = [ctx.lanczosResize(i, W // 2, H // 2) for i in imgs]
ctx.sync()
fns = {
    "adaptive": lambda k: ctx.AdaptiveSharpen(imgs[k], 0.5),
    "sharpen": lambda k: ctx.Sharpen(imgs[k], 0.5),
    "blur3": lambda k: ctx.blur3x3(imgs[k]),
    "ssim": lambda k: ctx.SSIM(imgs[k], other[k]),
    "ssimfast": lambda k: ctx.SSIMFast(imgs[k], other[k]),
    "msssim": lambda k: ctx.MSSSIM(imgs[k], other[k]),
    "resize_down": lambda k: ctx.lanczosResize(imgs[k], W // 2, H // 2),
    "resize_up": lambda k: ctx.lanczosResize(half[k], W, H),
}
for i in range(reps):
    fns[op](i % 3)
ctx.sync()
