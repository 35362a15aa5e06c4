#!/usr/bin/env python3
"""resize_fused_dense_kernel against the oracle: a plan reaches it only after the matrix kernel has handed its images back in
two cool-downs running (~66 calls with the same tables), so this keeps the SIZE fixed for a few hundred calls and varies the
content -- ramps (tie-dense), ramps with translucent pixels and patches, noise, seams -- checking every call.  Prints which
kernels ran.  python tools/fuzz_resize_dense.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fennec_amd  # noqa: E402
from fennec_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
orc.build()
fails, it, t0 = 0, 0, time.time()
seen = {}
while time.time() - t0 < budget:
    dw, dh = int(rng.integers(70, 700)), int(rng.integers(40, 400))
    w, h = 2 * dw, 2 * dh
    ctx = fennec_amd.Context(0)                       # a fresh ctx: its plans start without history
    ramp = synth.large_photo(w, h, int(rng.integers(100)))
    want_ramp = orc.lanczos_resize(ramp, dw, dh, procs=8)
    for k in range(int(rng.integers(140, 260))):
        kind = 0 if k < 70 else int(rng.integers(0, 5))        # the first 70 calls: ramps only (the plan must see dense retries)
        if kind == 0:
            img, want, what = ramp, want_ramp, "ramp"
        else:
            img = ramp.copy()
            if kind == 1:
                img[int(rng.integers(0, h)), int(rng.integers(0, w)), 3] = int(rng.integers(0, 255)); what = "ramp, one translucent px"
            elif kind == 2:
                y0, x0 = int(rng.integers(0, h)), int(rng.integers(0, w))
                img[y0:y0 + int(rng.integers(1, h // 3 + 2)), x0:x0 + int(rng.integers(1, w // 3 + 2)), 3] = int(rng.integers(0, 256)); what = "ramp, translucent patch"
            elif kind == 3:
                img = synth.noise_image(w, h, int(rng.integers(1 << 30)), alpha=True); img[..., 3] = 255; what = "noise"
            else:
                n = synth.noise_image(w, h, int(rng.integers(1 << 30)), alpha=True); n[..., 3] = 255
                img[:, : w // 2] = n[:, : w // 2]; what = "noise beside ramp"
            want = orc.lanczos_resize(img, dw, dh, procs=8)
        got = ctx.lanczosResize(img, dw, dh)
        route = ctx.last_kernel(fennec_amd.PROF_RESIZE)
        seen[route] = seen.get(route, 0) + 1
        it += 1
        if not np.array_equal(got, want):
            fails += 1
            d = np.argwhere(got != want)
            print(f"FAIL seed {seed} it {it}: {w}x{h} -> {dw}x{dh} call {k} ({what}, {route}): {len(d)} bytes differ, first at {d[0].tolist()}", flush=True)
    ctx.close()
print(f"{it} calls in {time.time() - t0:.0f} s, seed {seed}: {fails} failures; kernels: {seen}")
