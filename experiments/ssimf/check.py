#!/usr/bin/env python3
"""r6: the fp32-moment SSIM kernel (windowed_ssim_march2f_kernel) against the fp64 one: value difference and kernel time.
python experiments/ssimf/check.py [pairs]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fennec_amd  # noqa: E402
from fennec_amd import synth  # noqa: E402

ctx = fennec_amd.Context(0)
rng = np.random.default_rng(2)


def pairs(W, H):
    a = torch.from_numpy(synth.large_photo(W, H, 1)).cuda()
    yield "ramp/adaptive", a, ctx.AdaptiveSharpen(a, 0.5)
    yield "ramp/blur2", a, ctx.GaussianBlur(a, 2.0)
    n = torch.from_numpy(synth.noise_image(W, H, 3)).cuda()
    yield "noise/blur1.2", n, ctx.GaussianBlur(n, 1.2)
    yield "unrelated", a, n
    yield "identical", a, a.clone()
    y, x = np.mgrid[0:H, 0:W]
    ph = np.clip(np.stack([128 + 100 * np.sin(x / 97.0) * np.cos(y / 61.0) + rng.normal(0, 4, x.shape) for _ in range(3)] + [np.full(x.shape, 255.0)], -1), 0, 255).astype(np.uint8)
    p = torch.from_numpy(ph).cuda()
    yield "photo/adaptive", p, ctx.AdaptiveSharpen(p, 0.5)
    yield "photo/blur2", p, ctx.GaussianBlur(p, 2.0)
    br = torch.from_numpy(np.ascontiguousarray(np.clip(ph.astype(int) // 8 + 224, 0, 255).astype(np.uint8))).cuda()
    br[..., 3] = 255
    yield "bright/blur1", br, ctx.GaussianBlur(br, 1.0)
    fl = torch.full((H, W, 4), 250, dtype=torch.uint8, device="cuda")
    fb = fl.clone()
    fb[..., :3] = torch.from_numpy(np.clip(250 + rng.integers(-6, 6, (H, W, 3)), 0, 255).astype(np.uint8)).cuda()
    yield "flat250/noisy", fl, fb


for (W, H) in [(7680, 4320), (3840, 2160), (2600, 1700)]:
    for name, a, b in pairs(W, H):
        ctx.sync()
        out = {}
        for fast in (False, True):
            ctx.set_ssim_mode(fast)
            for _ in range(3):
                ctx.SSIM(a, b)
            ctx.profile(fennec_amd.PROF_SSIM)
            ms = []
            for _ in range(8):
                v = ctx.SSIM(a, b)
                ms.append(ctx.kernel_ms())
            ctx.profile(0)
            out[fast] = (v, np.mean(ms) * 1e3, ctx.last_kernel(fennec_amd.PROF_SSIM))
        ctx.set_ssim_mode(False)
        print(f"{W}x{H} {name:16s} exact {out[False][0]:.12f} {out[False][1]:7.1f} us | fast {out[True][0]:.12f} {out[True][1]:7.1f} us  "
              f"delta {out[True][0] - out[False][0]:+.2e}  [{out[True][2]}]", flush=True)
    if len(sys.argv) > 1 and sys.argv[1] == "8k":
        break
