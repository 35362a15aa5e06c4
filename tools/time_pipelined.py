#!/usr/bin/env python3
"""Two contexts, each with a FULL batch and its own destinations, enqueued alternately: step s+1 is queued
before step s's results are fetched, so the small tail kernels of one step run under the blur kernel of the
next.  python tools/time_pipelined.py [B] [depth]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fennec_amd  # noqa: E402
from fennec_amd import synth  # noqa: E402

W, H = 3840, 2160
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
D = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ctxs = [fennec_amd.Context(0) for _ in range(D)]
srcs = [torch.from_numpy(synth.large_photo(W, H, k)).cuda() for k in range(B)]
dsts = [[torch.empty_like(s) for s in srcs] for _ in range(D)]
torch.cuda.synchronize()
plans = [c.plan_blur_ssim_fast_batch(srcs, 2.0, outs=d) for c, d in zip(ctxs, dsts)]


def serial(n):
    for _ in range(n):
        plans[0].enqueue()
        plans[0].fetch()


def pipelined(n):
    for s in range(n + D - 1):
        if s < n:
            plans[s % D].enqueue()
        if s >= D - 1:
            plans[(s - D + 1) % D].fetch()


kms = []


def staggered(n):
    """step s+1 is enqueued when step s's BLUR kernel has finished (host wait on its end event), so two blur
    kernels never share the GPU; only a step's tail runs under the next blur"""
    plans[0].enqueue()
    for s in range(n):
        kms.append(ctxs[s % D].kernel_ms())
        if s + 1 < n:
            plans[(s + 1) % D].enqueue()
        plans[s % D].fetch()


for c in ctxs:
    c.profile(True)
for name, fn in (("serial", serial), ("pipelined", pipelined), ("staggered", staggered), ("serial", serial),
                 ("pipelined", pipelined), ("staggered", staggered)):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        fn(4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(40)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 40
    print(f"B={B} D={D} {name:10s} {dt * 1e3:7.4f} ms/step {W * H * B / dt / 1e6:9.0f} MP/s"
          + (f"  blur kernel {sum(kms[-40:]) / 40:.4f} ms" if name == "staggered" else ""), flush=True)
