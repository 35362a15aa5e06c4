#!/usr/bin/env python3
"""One 4K GaussianBlur call at a time (n = 1): the kernel's HIP-event time and the call's wall time."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fennec_amd
from fennec_amd import synth
ctx = fennec_amd.Context(0)
for (w, h) in ((3840, 2160), (1920, 1080), (7680, 4320)):
    src = torch.from_numpy(synth.large_photo(w, h, 1)).cuda()
    dst = torch.empty_like(src)
    torch.cuda.synchronize()
    for exact in (False, True):
        p = ctx.plan_blur_batch([src], 2.0, outs=[dst], exact=exact)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.5:
            p.run()
        ctx.sync()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ext = torch.cuda.ExternalStream(ctx.stream)
        e0.record(ext)
        for _ in range(200):
            p.run()
        e1.record(ext)
        e1.synchronize()
        print(f"{w}x{h} {'exact' if exact else 'fast '} {e0.elapsed_time(e1) / 200 * 1e3:7.2f} us per call back to back")
