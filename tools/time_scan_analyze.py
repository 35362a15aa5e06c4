#!/usr/bin/env python3
"""isOpaque and Analyze one call at a time at 4K through the Python binding (run it under `rocprofv3 --kernel-trace --stats` for the
kernels' own durations: scan_flags_direct_kernel, analyze_one_kernel, analyze_tail_kernel).  python tools/time_scan_analyze.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fennec_amd
from fennec_amd import synth
ctx = fennec_amd.Context(0)
imgs = [torch.from_numpy(synth.large_photo(3840, 2160, k)).cuda() for k in range(4)]
torch.cuda.synchronize()
for name, fn in (("isOpaque", lambda k: ctx.isOpaque(imgs[k])), ("Analyze", lambda k: ctx.Analyze(imgs[k]))):
    for i in range(50): fn(i % 4)
    t0 = time.perf_counter()
    for i in range(200): fn(i % 4)
    print(name, (time.perf_counter() - t0) / 200 * 1e6, "us/call", flush=True)
