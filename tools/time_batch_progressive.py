#!/usr/bin/env python3
"""fennec_CompressBatchJPEG (the C++ pool, no host codec) over 4K PROGRESSIVE files: the scans are entropy-decoded on the workers'
host threads (jpeg_prog.cpp), everything behind them runs on the device.  python tools/time_batch_progressive.py [items]"""
import ctypes as C
import io
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fennec_amd as fa  # noqa: E402
from fennec_amd import batch as fb, synth  # noqa: E402


def progressive(img, q):
    from PIL import Image
    b = io.BytesIO()
    Image.fromarray(np.ascontiguousarray(img[..., :3]), "RGB").save(b, "JPEG", quality=q, subsampling=2, progressive=True)
    return b.getvalue()


n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
base = [progressive(s, 92) for s in synth.large_photo_batch(3840, 2160, range(16))]
files = (base * ((n + 15) // 16))[:n]
print(f"{n} progressive files of {np.mean([len(f) for f in files]) / 1e6:.2f} MB (synth.large_photo, libjpeg q=92 4:2:0)", flush=True)
L = fa.load_library()
arrs = [np.frombuffer(f, dtype=np.uint8) for f in files]
srcs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
sizes = (C.c_size_t * n)(*[len(f) for f in files])
bufs = [np.empty(max(4096, 2 * len(f)), dtype=np.uint8) for f in files]
outs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
caps = (C.c_size_t * n)(*[b.size for b in bufs])
res = (fa.NativeBatchResult * n)()
for nw in (1, 4, 16, 32, 64):
    best = 0.0
    for rep in range(3):
        t = time.perf_counter()
        rc = L.fennec_CompressBatchJPEG(0, nw, n, srcs, sizes, 0.94, outs, caps, res, None, None, None)
        dt = time.perf_counter() - t
        assert rc == 0 and not any(r.failed for r in res)
        if rep:
            best = max(best, n / dt)
    print(f"C++ pool, {nw:2d} workers: {best:8.1f} images/s (best of 2)", flush=True)
t = time.perf_counter()
for f in files[:16]:
    fb.pillow_decode(f)
print(f"host codec (libjpeg-turbo through Pillow), one thread, decode only: {16 / (time.perf_counter() - t):8.1f} images/s", flush=True)
