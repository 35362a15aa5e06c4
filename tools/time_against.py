#!/usr/bin/env python3
"""PCIe-inclusive time of one step of the JPEG quality search's GPU stage (compress.go:45-74):
SSIMFast(prepared source, decoded candidate) with the candidate handed over as host NRGBA
(fnx_ssim_fast_against) or as the decoder's YCbCr planes (fnx_ssim_fast_against_ycbcr)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fennec_amd  # noqa: E402
from fennec_amd import synth  # noqa: E402

ctx = fennec_amd.Context(0)
src = synth.large_photo(3840, 2160, 0)
prep = ctx.ssim_fast_prepare(src)
for ratio, name in ((2, "4:2:0"), (1, "4:2:2"), (0, "4:4:4")):
    y, cb, cr = synth.rgb_to_ycbcr_planes(src, ratio)
    dec = ctx.ycbcrToNRGBA(y, cb, cr, ratio)
    for fn, label, nbytes in ((lambda: prep.against(dec), "NRGBA", dec.nbytes),
                              (lambda: prep.against_ycbcr(y, cb, cr, ratio), f"YCbCr {name}", y.nbytes + cb.nbytes + cr.nbytes)):
        fn(); fn()
        t0 = time.perf_counter()
        for _ in range(20):
            v = fn()
        dt = (time.perf_counter() - t0) / 20
        print(f"{label:12s} {nbytes / 1e6:6.1f} MB up  {dt * 1e3:7.3f} ms/call  ssim={v:.6f}")
