#!/usr/bin/env python3
"""Steady-state time of the one-pass blur + SSIMFast step and of the plain blur (4K, 32 images): python tools/time_blur_kernel.py [exact]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fennec_amd
from fennec_amd import synth
W, H, B = 3840, 2160, 32
exact = "exact" in sys.argv[1:]
rnd = "random" in sys.argv[1:]
sigmas = [float(a[6:]) for a in sys.argv[1:] if a.startswith("sigma=")] or [2.0]
SIG = sigmas[0]
ctx = fennec_amd.Context(0)
srcs = [torch.randint(0, 256, (H, W, 4), dtype=torch.uint8, device="cuda") if rnd else torch.from_numpy(synth.large_photo(W, H, k)).cuda() for k in range(B)]
dsts = [torch.empty_like(s) for s in srcs]
torch.cuda.synchronize()
one = ctx.plan_blur_ssim_fast_batch(srcs, SIG, outs=dsts, exact=exact)
pb = ctx.plan_blur_batch(srcs, SIG, outs=dsts, exact=exact)
for name, fn in (("one-pass", one.run), ("blur", pb.run)):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.0:
        fn()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(100):
        fn()
    ctx.sync()
    dt = (time.perf_counter() - t0) / 100
    print(f"{name:9s} sigma {SIG} {'exact' if exact else 'fast'} {dt / B * 1e6:8.2f} us/img  {W * H * B / dt / 1e6:10.0f} MP/s")
