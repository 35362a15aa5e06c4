#!/usr/bin/env python3
"""Kernel time of the JPEG quantisation round trip (FNX_PROF_JPEG brackets the block kernel) and wall time of the
whole quality search.  python tools/time_jpeg.py [W H]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fennec_amd  # noqa: E402
from fennec_amd import synth  # noqa: E402

sizes = [(int(sys.argv[1]), int(sys.argv[2]))] if len(sys.argv) > 2 else [(1920, 1080), (3840, 2160)]
ctx = fennec_amd.Context(0)
for W, H in sizes:
    img = torch.from_numpy(synth.large_photo(W, H, 1)).cuda()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2:
        ctx.jpeg_roundtrip(img, 60)
        ctx.sync()
    ctx.profile(fennec_amd.PROF_JPEG)
    ms = []
    for q in (30, 60, 92, 60, 30, 92):
        ctx.jpeg_roundtrip(img, q)
        ms.append(ctx.kernel_ms())
    ctx.profile(0)
    m = float(np.mean(ms))
    blocks = ((W + 15) // 16) * ((H + 15) // 16) * 6
    print(f"{W}x{H}: block kernel {m * 1e3:7.1f} us ({blocks} blocks, {W * H * 1.5 * 2 / m / 1e6:6.0f} GB/s of plane traffic)", flush=True)
    for call, label in ((lambda: ctx.jpeg_roundtrip(img, 60), "round trip (ycc + blocks + to NRGBA)"),
                        (lambda: ctx.jpeg_quality_search(img, 0.94), "quality search, target 0.94"),
                        (lambda: ctx.jpeg_encode(img, 60), "encode to a file (q=60, incl. 2 syncs + D2H)"),
                        (lambda: ctx.jpeg_compress(img, 0.94), "compress = search + encode at the winner")):
        call(); ctx.sync()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            r = call()
        ctx.sync()
        dt = (time.perf_counter() - t0) / n
        extra = ""
        if isinstance(r, tuple) and isinstance(r[0], bytes):
            extra = f" -> {len(r[0])} bytes q={r[1]} ssim={r[2]:.5f} steps={r[3]}"
        elif isinstance(r, tuple):
            extra = f" -> q={r[0]} ssim={r[1]:.5f} steps={r[2]}"
        elif isinstance(r, bytes):
            extra = f" -> {len(r)} bytes"
        print(f"    {label:40s} {dt * 1e3:8.3f} ms{extra}", flush=True)
