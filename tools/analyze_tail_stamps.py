#!/usr/bin/env python3
"""Phase stamps of analyze_tail_kernel's contrast workgroup (a DEVELOP build only: make -C fennec_amd/csrc BUILD=build_dev
OUT=../libfennec_hip_dev.so DEVELOP=1; run with FENNEC_HIP_LIB=.../libfennec_hip_dev.so FNX_AN_STAMPS=1): the stamps come back in
the edge_total field.  This is how round 5 found the 20 us single-CU contrast stage."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fennec_amd
from fennec_amd import synth
ctx = fennec_amd.Context(0)
imgs = [torch.from_numpy(synth.large_photo(3840, 2160, k)).cuda() for k in range(4)]
torch.cuda.synchronize()
for i in range(100): ctx.Analyze(imgs[i % 4])
for i in range(5):
    r = ctx.analyze_raw(imgs[i % 4])
    pk = int(r["edge_total"])
    print("stamps (us since tail start): bright loaded %.2f, sum %.2f, contrast loaded+computed %.2f, sum %.2f, stores acked %.2f" % tuple(((pk >> (12 * k)) & 0xfff) / 100.0 for k in range(5)))
