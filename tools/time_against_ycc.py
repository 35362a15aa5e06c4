#!/usr/bin/env python3
"""SSIMFast(prepared 4K reference, candidate planes on the device), repeated -- for rocprofv3 --stats: the candidate plane's
kernels alone (box_tiled_ycc_kernel, or ycbcr_to_nrgba_kernel + box_tiled_kernel with FNX_BOX_YCC=0).
python tools/time_against_ycc.py [ratio] [reps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fennec_amd  # noqa: E402
from fennec_amd import synth  # noqa: E402

ratio = int(sys.argv[1]) if len(sys.argv) > 1 else 2
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
ctx = fennec_amd.Context(0)
src = synth.large_photo(3840, 2160, 6)
prep = ctx.ssim_fast_prepare(src)
y, cb, cr = (torch.from_numpy(p).cuda() for p in synth.rgb_to_ycbcr_planes(src, ratio))
for _ in range(5):
    v = prep.against_ycbcr(y, cb, cr, ratio)
t = time.perf_counter()
for _ in range(reps):
    v = prep.against_ycbcr(y, cb, cr, ratio)
print(f"ratio {ratio}: SSIMFast {v:.9f}  {1e6 * (time.perf_counter() - t) / reps:.1f} us per call (host time, one call at a time)")
