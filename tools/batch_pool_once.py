#!/usr/bin/env python3
"""fennec_CompressBatchJPEG (the C++ pool) once over N 4K JPEG files with W workers -- for rocprofv3 --kernel-trace
(tools/trace_overlap.py reads the trace).  python tools/batch_pool_once.py [items] [workers] [reps]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fennec_amd as fa  # noqa: E402
from fennec_amd import batch as fb, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
nw = int(sys.argv[2]) if len(sys.argv) > 2 else 12
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
base = [fb.pillow_encode(s, 92) for s in synth.large_photo_batch(3840, 2160, range(8))]
files = (base * ((n + 7) // 8))[:n]
L = fa.load_library()
arrs = [np.frombuffer(f, dtype=np.uint8) for f in files]
srcs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
sizes = (C.c_size_t * n)(*[len(f) for f in files])
bufs = [np.empty(max(4096, 2 * len(f)), dtype=np.uint8) for f in files]
outs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
caps = (C.c_size_t * n)(*[b.size for b in bufs])
res = (fa.NativeBatchResult * n)()
for rep in range(reps):
    t = time.perf_counter()
    rc = L.fennec_CompressBatchJPEG(0, nw, n, srcs, sizes, 0.94, outs, caps, res, None, None, None)
    dt = time.perf_counter() - t
    assert rc == 0 and not any(r.failed for r in res)
    print(f"{nw} workers, {n} files of {np.mean([len(f) for f in files]) / 1e6:.2f} MB: {n / dt:.1f} images/s", flush=True)
