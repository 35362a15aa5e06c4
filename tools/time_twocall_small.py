#!/usr/bin/env python3
"""Do SSIMFast's box kernels hit the Infinity Cache when they follow the blur of the same FEW images?
python tools/time_twocall_small.py B  -> per-image time of blur(B) + ssim_fast(B) over 32 images in groups of B"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fennec_amd
from fennec_amd import synth
W, H, N = 3840, 2160, 32
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ctx = fennec_amd.Context(0)
srcs = [torch.from_numpy(synth.large_photo(W, H, k)).cuda() for k in range(N)]
dsts = [torch.empty_like(s) for s in srcs]
torch.cuda.synchronize()
groups = [list(range(i, i + B)) for i in range(0, N, B)]
pb = [ctx.plan_blur_batch([srcs[i] for i in g], 2.0, outs=[dsts[i] for i in g]) for g in groups]
ps = [ctx.plan_ssim_fast_batch([srcs[i] for i in g], [dsts[i] for i in g]) for g in groups]
def step():
    out = []
    for k, (b, s) in enumerate(zip(pb, ps)):
        b.run()
        s.enqueue()
        if k >= 3:
            out.append(ps[k - 3].fetch())     # the ctx holds at most four unfetched batches
    for s in ps[max(0, len(ps) - 3):]:
        out.append(s.fetch())
    return out
t0 = time.perf_counter()
while time.perf_counter() - t0 < 1.0:
    step()
ctx.sync()
t0 = time.perf_counter()
for _ in range(30):
    step()
ctx.sync()
dt = (time.perf_counter() - t0) / 30
print(f"groups of {B}: {dt / N * 1e6:.2f} us per image, {W * H * N / dt / 1e6:.0f} MP/s")
