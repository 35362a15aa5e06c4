#!/usr/bin/env python3
"""Kernel times of lanczosResize's two passes (HIP events inside the library, FNX_PROF_RESIZE).
python tools/time_resize.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fennec_amd  # noqa: E402
from fennec_amd import synth  # noqa: E402

ctx = fennec_amd.Context(0)
CASES = [(3840, 2160, 1920, 1080), (1920, 1080, 3840, 2160), (7680, 4320, 3840, 2160), (3840, 2160, 1280, 720),
         (3840, 2160, 2560, 1440)]
if os.environ.get("FNX_TR_CASES"):          # e.g. "0" or "0,1": PMC passes of one shape
    CASES = [CASES[int(i)] for i in os.environ["FNX_TR_CASES"].split(",")]
for (W, H, DW, DH) in CASES:
    imgs = [torch.from_numpy(synth.large_photo(W, H, k)).cuda() for k in range(3)]
    if len(sys.argv) > 1 and sys.argv[1] == "soft":       # blurred ramps: no exact ties, the guard decides nearly everything
        imgs = [ctx.GaussianBlur(ctx.GaussianBlur(torch.from_numpy(synth.noise_image(W, H, 5 + k)).cuda(), 2.0), 1.2) for k in range(3)]
    ctx.sync()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2:
        ctx.lanczosResize(imgs[0], DW, DH)
        ctx.sync()
    ctx.profile(fennec_amd.PROF_RESIZE)
    hs, vs = [], []
    t0 = time.perf_counter()
    for k in range(9):
        ctx.lanczosResize(imgs[k % 3], DW, DH)
        hs.append(ctx.kernel_ms())
        try:                                              # two launches (resizeH, resizeV) -- or one, the fused kernel
            vs.append(ctx.kernel_ms())
        except fennec_amd.FennecError:
            pass
    ctx.profile(0)
    S = 4.0 * (W * H + DW * DH)
    tot = float(np.mean(hs) + (np.mean(vs) if vs else 0.0))
    parts = f"H {np.mean(hs) * 1e3:7.1f} us  V {np.mean(vs) * 1e3:7.1f} us" if vs else "one launch (fused H + V)      "
    print(f"{W}x{H} -> {DW}x{DH}: {parts}  total {tot * 1e3:7.1f} us  "
          f"{S / tot / 1e6:7.0f} GB/s algorithmic ({S / tot / 1e6 / 8000:.3f} of 8 TB/s)", flush=True)
