#!/usr/bin/env python3
"""The fp32-moment SSIM kernel alone on 8K pairs (HIP events inside the library): python experiments/config4/ssim_alone.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fennec_amd  # noqa: E402
from fennec_amd import synth  # noqa: E402

ctx = fennec_amd.Context(0)
ctx.set_ssim_mode(True)
W, H = 7680, 4320
imgs = [torch.from_numpy(im).cuda() for im in synth.large_photo_batch(W, H, range(3))]
sh = [ctx.AdaptiveSharpen(im, 0.5) for im in imgs]
for _ in range(300):
    ctx.SSIM(imgs[0], sh[0])
ctx.profile(fennec_amd.PROF_SSIM)
ms, vals = [], []
for k in range(30):
    vals.append(ctx.SSIM(imgs[k % 3], sh[k % 3]))
    ms.append(ctx.kernel_ms())
ctx.profile(0)
print(f"SSIM fast 8K kernel {np.mean(ms) * 1e3:7.1f} us (min {np.min(ms) * 1e3:.1f})  [{ctx.last_kernel(fennec_amd.PROF_SSIM)}]  value {vals[0]:.12f}")
