#!/usr/bin/env python3
"""20 full-resolution SSIM calls of one 8K pair in the mode given (exact | fast): the command tools/pmc.sh profiles."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fennec_amd  # noqa: E402
from fennec_amd import synth  # noqa: E402

ctx = fennec_amd.Context(0)
a = torch.from_numpy(synth.large_photo(7680, 4320, 1)).cuda()
b = ctx.AdaptiveSharpen(a, 0.5)
ctx.set_ssim_mode(len(sys.argv) > 1 and sys.argv[1] == "fast")
for _ in range(20):
    v = ctx.SSIM(a, b)
print(v, ctx.last_kernel(fennec_amd.PROF_SSIM))
