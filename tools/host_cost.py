#!/usr/bin/env python3
"""Host-side cost of the per-image calls of config 3 (tiny images: the kernels are negligible)."""
import os, sys, time, threading
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fennec_amd
from fennec_amd import synth

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 256)
N = 200
for nthr in (1, 2, 4):
    ctxs = [fennec_amd.Context(0) for _ in range(nthr)]
    streams = [torch.cuda.Stream() for _ in range(nthr)]
    img = torch.from_numpy(synth.large_photo(W, H, 1)).cuda()
    def run(k):
        with torch.cuda.stream(streams[k]):
            c = ctxs[k]
            pend = 0
            for _ in range(N):
                small = c.lanczosResize(img, W // 2, H // 2)
                c.msssim_enqueue(img, small)
                pend += 1
                if pend > 3:
                    c.fetch_result(); pend -= 1
            while pend:
                c.fetch_result(); pend -= 1
    run(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ts = [threading.Thread(target=run, args=(k,)) for k in range(nthr)]
    [t.start() for t in ts]; [t.join() for t in ts]
    dt = time.perf_counter() - t0
    print(f"{W}x{H} threads={nthr}: {dt / (N * nthr) * 1e6:.1f} us per image (aggregate), {dt / N * 1e6:.1f} us per image per thread", flush=True)
c = ctxs[0]
t0 = time.perf_counter()
for _ in range(N):
    small = c.lanczosResize(img, W // 2, H // 2)
c.sync()
print(f"lanczosResize alone (async): {(time.perf_counter() - t0) / N * 1e6:.1f} us", flush=True)
t0 = time.perf_counter()
for _ in range(N):
    c.MSSSIM(img, small)
print(f"MSSSIM alone (sync): {(time.perf_counter() - t0) / N * 1e6:.1f} us", flush=True)
