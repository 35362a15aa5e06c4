#!/usr/bin/env python3
"""fnx_gaussian_blur_batch + fnx_ssim_fast_batch with and without FNX_BLUR_KEEP_BOX_SUMS, and the one-pass entry, per batch size:
python tools/time_twocall_keep.py [W H]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fennec_amd  # noqa: E402
from fennec_amd import synth  # noqa: E402

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
ctx = fennec_amd.Context(0)
for B in (1, 2, 3, 4, 6, 8, 16, 32):
    srcs = [torch.from_numpy(synth.large_photo(W, H, k)).cuda() for k in range(B)]
    dsts = [torch.empty_like(s) for s in srcs]
    torch.cuda.synchronize()
    plain = ctx.plan_blur_batch(srcs, 2.0, outs=dsts)
    keep = ctx.plan_blur_batch(srcs, 2.0, outs=dsts, keep_box_sums=True)
    score = ctx.plan_ssim_fast_batch(srcs, dsts)
    one = ctx.plan_blur_ssim_fast_batch(srcs, 2.0, outs=dsts)

    def two(p):
        p.run()
        return score.run()

    res = {}
    for name, fn in (("two calls", lambda: two(plain)), ("keep", lambda: two(keep)), ("one pass", one.run)) * 2:
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.25:
            fn()
        t0 = time.perf_counter()
        for _ in range(40):
            fn()
        res[name] = (time.perf_counter() - t0) / 40
    print(f"{W}x{H} B={B:2d}: " + "  ".join(f"{k} {v * 1e6:8.1f} us ({v / B * 1e6:6.1f} per image)" for k, v in res.items()), flush=True)
    del srcs, dsts
