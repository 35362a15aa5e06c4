#!/usr/bin/env python3
"""One-pass (fnx_gaussian_blur_ssim_fast_batch) vs two-call timing at any image size:
python tools/time_onepass.py W H B [sigma]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fennec_amd  # noqa: E402
from fennec_amd import synth  # noqa: E402

W, H, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
sigma = float(sys.argv[4]) if len(sys.argv) > 4 else 2.0
ctx = fennec_amd.Context(0)
srcs = [torch.from_numpy(synth.large_photo(W, H, k)).cuda() for k in range(B)]
dsts = [torch.empty_like(s) for s in srcs]
torch.cuda.synchronize()
one = ctx.plan_blur_ssim_fast_batch(srcs, sigma, outs=dsts)
onex = ctx.plan_blur_ssim_fast_batch(srcs, sigma, outs=dsts, exact=True)
pb = ctx.plan_blur_batch(srcs, sigma, outs=dsts)
ps = ctx.plan_ssim_fast_batch(srcs, dsts)


def two():
    pb.run()
    return ps.run()


for name, fn in (("one-pass", one.run), ("two-call", two), ("one-pass", one.run), ("two-call", two),
                 ("1p-exact", onex.run), ("1p-exact", onex.run)):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:      # clock ramp (see bench.py PREWARM_S)
        v = fn().copy()
    t0 = time.perf_counter()
    for _ in range(20):
        fn()
    dt = (time.perf_counter() - t0) / 20
    print(f"{W}x{H} B={B} {name:9s} {dt / B * 1e6:8.2f} us/img  {W * H * B / dt / 1e6:10.0f} MP/s  ssim[0]={v[0]:.12f}")
