#!/usr/bin/env python3
"""Kernel time of gaussianBlur3x3 / Sharpen / AdaptiveSharpen (HIP events inside the library, FNX_PROF_FX).
python tools/time_fx.py [W H]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fennec_amd  # noqa: E402
from fennec_amd import synth  # noqa: E402

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (7680, 4320)
ctx = fennec_amd.Context(0)
imgs = [torch.from_numpy(synth.large_photo(W, H, k)).cuda() for k in range(3)]
soft = [ctx.GaussianBlur(ctx.GaussianBlur(i, 2.0), 1.2) for i in imgs]
ctx.sync()
S = 4.0 * W * H
for label, pool in (("photo", imgs), ("soft", soft)):
    for name, fn in (("blur3x3", lambda a: ctx.blur3x3(a)), ("Sharpen 0.5", lambda a: ctx.Sharpen(a, 0.5)),
                     ("AdaptiveSharpen 0.5", lambda a: ctx.AdaptiveSharpen(a, 0.5)),
                     ("AdaptiveSharpen 0.25 (ties)", lambda a: ctx.AdaptiveSharpen(a, 0.25))):
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.2:
            fn(pool[0])
            ctx.sync()
        ctx.profile(fennec_amd.PROF_FX)
        ms = []
        for k in range(9):
            fn(pool[k % 3])
            ms.append(ctx.kernel_ms())
        ctx.profile(0)
        m = float(np.mean(ms))
        print(f"{label:6s} {name:28s} {m * 1e3:8.1f} us  {2 * S / m / 1e6:8.0f} GB/s  {2 * S / m / 1e6 / 8000:.3f} of 8 TB/s", flush=True)
