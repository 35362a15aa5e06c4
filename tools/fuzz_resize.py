#!/usr/bin/env python3
"""Randomised lanczosResize sweep against the oracle with the matrix kernel taking every table it covers
(FNX_RESIZE_MFMA=2 unless set): python tools/fuzz_resize.py [seconds] [seed].  Sizes 40..2600, ratios mostly 0.4..2.4 per
axis (independent), content: opaque noise, structured photo-like, SURVEY 8(d)'s ramp, few-colour ties, translucent patches
and single translucent pixels, mixtures.  Every failure prints the reproducing seed and iteration."""
import os
import sys
import time

import numpy as np

os.environ.setdefault("FNX_RESIZE_MFMA", "2")
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
    kind = int(rng.integers(0, 7))
    noise = synth.noise_image(w, h, int(rng.integers(1 << 30)), alpha=True)
    noise[..., 3] = 255
    if kind == 0:
        return noise, "noise"
    y, x = np.mgrid[0:h, 0:w]
    photo = noise.copy()
    base = 127 + 90 * np.sin(x / float(rng.uniform(5, 60))) * np.cos(y / float(rng.uniform(5, 60))) + 30 * ((x // 50 + y // 40) % 2)
    for c in range(3):
        photo[..., c] = np.clip(base + 15 * c + rng.normal(0, float(rng.uniform(0, 8)), (h, w)), 0, 255).astype(np.uint8)
    if kind == 1:
        return photo, "photo"
    if kind == 2:
        return synth.large_photo(w, h, int(rng.integers(100))), "ramp"
    if kind == 3:
        photo[..., :3] &= 0xF0
        return photo, "few colours"
    if kind == 4:
        for _ in range(int(rng.integers(1, 4))):
            y0, x0 = int(rng.integers(0, h)), int(rng.integers(0, w))
            photo[y0:y0 + int(rng.integers(1, h // 3 + 2)), x0:x0 + int(rng.integers(1, w // 3 + 2)), 3] = int(rng.integers(0, 255))
        return photo, "translucent patches"
    if kind == 5:
        photo[int(rng.integers(0, h)), int(rng.integers(0, w)), 3] = 254
        return photo, "one translucent px"
    ramp = synth.large_photo(w, h, int(rng.integers(100)))
    photo[: h // 2, : w // 2] = ramp[: h // 2, : w // 2]
    return photo, "ramp corner"


while time.time() - t0 < budget:
    it += 1
    w, h = int(rng.integers(40, 2600)), int(rng.integers(40, 1500))
    if w * h > 2_500_000:
        h = max(40, 2_500_000 // w)
    rx = float(rng.uniform(0.4, 2.4)) if rng.integers(8) else float(rng.choice([1.0, 2.0, 0.5, 1.5, 3.0, 0.25]))
    ry = rx if rng.integers(3) else float(rng.uniform(0.4, 2.4))
    dw, dh = max(1, int(round(w / rx))), max(1, int(round(h / ry)))
    img, what = content(w, h)
    got = ctx.lanczosResize(img, dw, dh)
    want = orc.lanczos_resize(img, dw, dh, procs=16)
    if not np.array_equal(got, want):
        fails += 1
        d = np.argwhere(got != want)
        print(f"FAIL seed {seed} it {it}: {w}x{h} -> {dw}x{dh} ({what}): {len(d)} bytes differ, first at {d[0].tolist()}", flush=True)
print(f"{it} iterations in {time.time() - t0:.0f} s, seed {seed}, FNX_RESIZE_MFMA={os.environ['FNX_RESIZE_MFMA']}: {fails} failures")
