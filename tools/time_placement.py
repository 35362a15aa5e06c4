#!/usr/bin/env python3
"""Does the placement of the image buffers matter to the one-pass kernel?  B sources and B destinations
carved from ONE allocation at a stride of S + pad bytes (torch's allocator rounds a 4K image to exactly
32 MiB, so separately allocated buffers sit at 2^25-byte strides):
python tools/time_placement.py [B]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fennec_amd  # noqa: E402
from fennec_amd import synth  # noqa: E402

W, H = 3840, 2160
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
S = W * H * 4
ctx = fennec_amd.Context(0)
host = [torch.from_numpy(synth.large_photo(W, H, k)) for k in range(4)]
for pad in (-1, 0, 4096, 65536 + 256, (1 << 20) + 4096, 32 * 1024 * 1024 - S, 32 * 1024 * 1024 - S + 8192):
    if pad < 0:       # separately allocated tensors (what bench.py does)
        srcs = [host[k % 4].cuda() for k in range(B)]
        dsts = [torch.empty_like(s) for s in srcs]
        label = "separate allocations"
    else:
        stride = S + pad
        pool = torch.empty(2 * B * stride + 4096, dtype=torch.uint8, device="cuda")
        view = lambda i: pool[i * stride:i * stride + S].view(H, W, 4)
        srcs = [view(2 * k) for k in range(B)]
        dsts = [view(2 * k + 1) for k in range(B)]
        for k in range(B):
            srcs[k].copy_(host[k % 4])
        label = f"one pool, stride S + {pad}"
    torch.cuda.synchronize()
    plan = ctx.plan_blur_ssim_fast_batch(srcs, 2.0, outs=dsts)
    ctx.profile(True)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        plan.run()
    ks = []
    t0 = time.perf_counter()
    for _ in range(20):
        plan.enqueue()
        ks.append(ctx.kernel_ms())
        plan.fetch()
    dt = (time.perf_counter() - t0) / 20
    ctx.profile(False)
    print(f"B={B} {label:42s} step {dt * 1e3:7.4f} ms  kernel {sum(ks) / len(ks):7.4f} ms  {W * H * B / dt / 1e6:9.0f} MP/s", flush=True)
    del plan, srcs, dsts
    torch.cuda.empty_cache()
