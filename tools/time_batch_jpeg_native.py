#!/usr/bin/env python3
"""CompressBatch over 4K JPEG files with no host codec: fennec_CompressBatchJPEG (the C++ pool; the C call alone, then with the
python wrapper's buffer handling) and the python harness (compress_batch + fnx_jpeg_recompress per item) at several worker
counts.  python tools/time_batch_jpeg_native.py [items]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fennec_amd as fa  # noqa: E402
from fennec_amd import batch as fb, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
base = [fb.pillow_encode(s, 92) for s in synth.large_photo_batch(3840, 2160, range(16))]
files = (base * ((n + 15) // 16))[:n]
print(f"{n} files of {np.mean([len(f) for f in files]) / 1e6:.2f} MB (synth.large_photo, libjpeg q=92 4:2:0)", flush=True)
L = fa.load_library()
arrs = [np.frombuffer(f, dtype=np.uint8) for f in files]
srcs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
sizes = (C.c_size_t * n)(*[len(f) for f in files])
bufs = [np.empty(max(4096, 2 * len(f)), dtype=np.uint8) for f in files]
outs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
caps = (C.c_size_t * n)(*[b.size for b in bufs])
res = (fa.NativeBatchResult * n)()
for nw in (1, 2, 4, 8, 16):
    best = 0.0
    for rep in range(4):
        t = time.perf_counter()
        rc = L.fennec_CompressBatchJPEG(0, nw, n, srcs, sizes, 0.94, outs, caps, res, None, None, None)
        dt = time.perf_counter() - t
        assert rc == 0 and not any(r.failed for r in res)
        if rep:
            best = max(best, n / dt)
    t = time.perf_counter()
    fb.compress_batch_jpeg_native(files, 0.94, workers=nw)
    wrap = n / (time.perf_counter() - t)
    print(f"C++ pool, {nw:2d} workers: {best:8.1f} images/s (the call alone, best of 3); with the python wrapper {wrap:8.1f}", flush=True)

states = {}


def make_state(wid):
    if wid not in states:
        states[wid] = fa.Context(0)
    return states[wid]


for nw in (1, 2, 4, 8):
    work = fb.jpeg_item_work_device_all(files, 0.94)
    fb.compress_batch(n, work, make_state, workers=nw)
    t = time.perf_counter()
    fb.compress_batch(n, work, make_state, workers=nw)
    print(f"python harness, {nw:2d} workers: {n / (time.perf_counter() - t):8.1f} images/s", flush=True)
