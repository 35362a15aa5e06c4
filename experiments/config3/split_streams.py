#!/usr/bin/env python3
"""Config 3 with the resizes and the MSSSIMs on DIFFERENT streams (a feasibility probe for batched resize entry points):
stream 0 runs the 32 downscales and the 32 upscales of step k back to back (what two batched launches would do, minus their
fewer tails), four worker contexts run MSSSIM(A_i, up_i) (equal dims) of step k - 1 beside them.  Prints MP/s next to the
per-image flow bench.py uses (lanczosResize + msssim_enqueue per image on four contexts)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fennec_amd  # noqa: E402
from fennec_amd import synth  # noqa: E402

W, H, B, NW = 3840, 2160, 32, 4
imgs = [torch.from_numpy(im).cuda() for im in synth.large_photo_batch(W, H, range(B))]
torch.cuda.synchronize()
c0 = fennec_amd.Context(0)
workers = [fennec_amd.Context(0) for _ in range(NW)]
s0 = torch.cuda.Stream()
ws = [torch.cuda.Stream() for _ in range(NW)]
ups = [[None] * B, [None] * B]


def resizes(k):
    with torch.cuda.stream(s0):
        small = [c0.lanczosResize(imgs[i], W // 2, H // 2) for i in range(B)]
        ups[k & 1] = [c0.lanczosResize(small[i], W, H) for i in range(B)]
        ev = torch.cuda.Event()
        ev.record(s0)
    return ev, small


def scores(k, ev):
    out = [None] * B
    pend = [[] for _ in range(NW)]
    for w in range(NW):
        ws[w].wait_event(ev)
    for i in range(B):
        w = i % NW
        with torch.cuda.stream(ws[w]):
            workers[w].msssim_enqueue(imgs[i], ups[k & 1][i])
            pend[w].append(i)
            if len(pend[w]) > 3:
                out[pend[w].pop(0)] = workers[w].fetch_result()
    for w in range(NW):
        with torch.cuda.stream(ws[w]):
            while pend[w]:
                out[pend[w].pop(0)] = workers[w].fetch_result()
    return out


def run(steps):
    ev, keep = resizes(0)
    vals = None
    for k in range(1, steps + 1):
        ev2, keep2 = resizes(k)              # step k's resizes are queued before step k - 1's scores are fetched
        vals = scores(k - 1, ev)
        ev, keep = ev2, keep2
    vals = scores(steps, ev)
    torch.cuda.synchronize()
    return vals


t0 = time.perf_counter()
while time.perf_counter() - t0 < 1.5:
    run(2)
for trial in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    v = run(n)
    dt = time.perf_counter() - t0
    print(f"split streams: {(n + 1) * B * W * H / 1e6 / dt:9.0f} MP/s  ({dt / (n + 1) * 1e3:.3f} ms per step of {B}), MSSSIM[0] = {v[0]:.12f}", flush=True)

# the same arithmetic through the reference-shaped call (MSSSIM(A, small) upscales inside): must agree
small = c0.lanczosResize(imgs[0], W // 2, H // 2)
c0.msssim_enqueue(imgs[0], small)
print("reference-shaped MSSSIM[0] =", f"{c0.fetch_result():.12f}")
