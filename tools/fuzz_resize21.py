#!/usr/bin/env python3
"""lanczosResize at and around 2:1 against the oracle, on content that sends resize_fused_kernel to its exact loops (rounding
ties: SURVEY 8(d)'s ramp and linear ramps; ramps with translucent pixels and patches; ramp / noise mixtures) -- the 2:1 forms
of those loops (scalar weights, four outputs per lane) with their edge groups, odd widths and heights, lone last rows.
FNX_RESIZE_MFMA=0 unless set (the matrix kernel would take the opaque tiles first).  python tools/fuzz_resize21.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

os.environ.setdefault("FNX_RESIZE_MFMA", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fennec_amd  # noqa: E402
from fennec_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
orc.build()
ctx = fennec_amd.Context(0)
ctx.set_form("resize_mfma", os.environ["FNX_RESIZE_MFMA"])     # (r6: a per-ctx kernel-form selection; the variable is this tool's own)
fails, it, t0 = 0, 0, time.time()


def content(w, h):
    kind = int(rng.integers(0, 6))
    ramp = synth.large_photo(w, h, int(rng.integers(100)))
    if kind == 0:
        return ramp, "ramp"
    y, x = np.mgrid[0:h, 0:w]
    if kind == 1:                                   # linear ramps of random integer slopes: ties in both passes
        lin = np.empty((h, w, 4), np.uint8)
        for c in range(3):
            lin[..., c] = (int(rng.integers(0, 5)) * x + int(rng.integers(0, 5)) * y + int(rng.integers(0, 256))) % 256
        lin[..., 3] = 255
        return lin, "linear"
    if kind == 2:
        ramp[int(rng.integers(0, h)), int(rng.integers(0, w)), 3] = int(rng.integers(0, 255))
        return ramp, "ramp, one translucent px"
    if kind == 3:
        for _ in range(int(rng.integers(1, 4))):
            y0, x0 = int(rng.integers(0, h)), int(rng.integers(0, w))
            ramp[y0:y0 + int(rng.integers(1, h // 3 + 2)), x0:x0 + int(rng.integers(1, w // 3 + 2)), 3] = int(rng.integers(0, 256))
        return ramp, "ramp, translucent patches"
    noise = synth.noise_image(w, h, int(rng.integers(1 << 30)), alpha=True)
    noise[..., 3] = 255
    if kind == 4:
        noise[: h // 2] = ramp[: h // 2]
        return noise, "ramp over noise"
    noise[:, w // 3:] = ramp[:, w // 3:]
    return noise, "noise beside ramp"


while time.time() - t0 < budget:
    it += 1
    dw, dh = int(rng.integers(20, 1400)), int(rng.integers(20, 800))
    if dw * dh > 700_000:
        dh = max(20, 700_000 // dw)
    mode = int(rng.integers(0, 5))
    w, h = 2 * dw, 2 * dh                           # exactly 2:1 on both axes
    if mode == 1:
        w += int(rng.integers(-1, 2))               # nearly 2:1 in x (not uniform: the masked loops)
    elif mode == 2:
        h += int(rng.integers(-1, 2))
    elif mode == 3:
        h = max(8, int(round(dh * float(rng.uniform(1.3, 2.2)))))   # 2:1 in x only
    img, what = content(w, h)
    got = ctx.lanczosResize(img, dw, dh)
    want = orc.lanczos_resize(img, dw, dh, procs=16)
    if not np.array_equal(got, want):
        fails += 1
        d = np.argwhere(got != want)
        print(f"FAIL seed {seed} it {it}: {w}x{h} -> {dw}x{dh} ({what}): {len(d)} bytes differ, first at {d[0].tolist()}", flush=True)
print(f"{it} iterations in {time.time() - t0:.0f} s, seed {seed}, FNX_RESIZE_MFMA={os.environ['FNX_RESIZE_MFMA']}: {fails} failures")
