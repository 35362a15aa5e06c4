#!/usr/bin/env python3
"""Kernel time of windowed SSIM (HIP events inside the library, FNX_PROF_SSIM) at several plane sizes.
python tools/time_ssim.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fennec_amd  # noqa: E402
from fennec_amd import synth  # noqa: E402

ctx = fennec_amd.Context(0)
for (W, H) in [(7680, 4320), (3840, 2160), (1920, 1080), (512, 288)]:
    a = torch.from_numpy(synth.large_photo(W, H, 1)).cuda()
    b = ctx.AdaptiveSharpen(a, 0.5)
    ctx.sync()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        ctx.SSIM(a, b)
    ctx.profile(fennec_amd.PROF_SSIM)
    ms = []
    t0 = time.perf_counter()
    for _ in range(10):
        v = ctx.SSIM(a, b)
        try:
            ms.append(ctx.kernel_ms())
        except fennec_amd.FennecError:          # small planes take the tile kernels, which are not bracketed
            ms.append(float("nan"))
    wall = (time.perf_counter() - t0) / 10
    ctx.profile(0)
    win = (W - 8) * (H - 8)
    print(f"{W}x{H}: SSIM={v:.12f} kernel {np.mean(ms) * 1e3:8.1f} us (min {np.min(ms) * 1e3:.1f})  call {wall * 1e6:8.1f} us  "
          f"{win / np.mean(ms) / 1e6:.1f} Gwin/s", flush=True)
