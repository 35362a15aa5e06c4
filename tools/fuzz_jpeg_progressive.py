#!/usr/bin/env python3
"""Randomised sweep of the progressive JPEG route on a GPU box: python tools/fuzz_jpeg_progressive.py [seconds] [seed].
libjpeg's progressive files (Pillow) of random sizes, contents, qualities, subsamplings, optimised tables or not, grey, restart
intervals where the route takes them -- fnx_jpeg_decode against the oracle's decode bit for bit, the host decoder's coefficients
against the oracle's, every fifth file fnx_jpeg_recompress against decode + fnx_jpeg_compress, every third file damaged (either
both sides refuse, or both decode to the same image).  Every failure prints its reproducing case."""
import io
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fennec_amd  # noqa: E402
from fennec_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

ZIG = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56,
       57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ctx = fennec_amd.Context(0)
orc.build()
fails, runs = [], {}


def case(name, ok, desc):
    runs[name] = runs.get(name, 0) + 1
    if not ok:
        fails.append(f"{name}: {desc}")


def content(w, h):
    k = int(rng.integers(5))
    if k == 0:
        return synth.noise_image(w, h, int(rng.integers(1 << 30)), alpha=False)
    if k == 1:
        a = np.full((h, w, 4), 255, np.uint8)
        a[..., :3] = rng.integers(0, 256, 3)
        return a
    if k == 2:                                    # few colours, hard edges
        a = synth.large_photo(w, h, int(rng.integers(1000)))
        a[..., :3] &= 0xc0
        return a
    return synth.large_photo(w, h, int(rng.integers(1000)))


def pil(img, grey, **kw):
    from PIL import Image
    b = io.BytesIO()
    if kw.pop("cmyk", False):                     # four components (Adobe CMYK; `ycck`: the same scans relabelled YCbCrK)
        ycck = kw.pop("ycck", False)
        Image.fromarray(np.ascontiguousarray(img[..., :3]), "RGB").convert("CMYK").save(b, "JPEG", **kw)
        d = b.getvalue()
        i = d.index(b"Adobe")
        return d[:i + 11] + (b"\x02" if ycck else b"\x00") + d[i + 12:]
    if grey:
        Image.fromarray(np.ascontiguousarray(img[..., 1]), "L").save(b, "JPEG", **kw)
    else:
        Image.fromarray(np.ascontiguousarray(img[..., :3]), "RGB").save(b, "JPEG", **kw)
    return b.getvalue()


t0, it = time.time(), 0
while time.time() - t0 < budget:
    it += 1
    w, h = (int(rng.integers(1, 400)), int(rng.integers(1, 300))) if it % 4 else (int(rng.integers(400, 2600)), int(rng.integers(300, 1700)))
    grey = it % 9 == 0
    sub = int(rng.integers(3))
    kw = dict(quality=int(rng.integers(5, 101)), progressive=True, optimize=bool(rng.integers(2)))
    if not grey:
        kw["subsampling"] = sub
    if (grey or sub == 0) and it % 3 == 0:        # restart intervals where image/jpeg's count and T.81's agree
        kw["restart_marker_blocks"] = int(rng.integers(1, 40))
    if it % 11 == 0:                              # four components: any frame type goes the host's way
        grey = False
        kw = dict(quality=kw["quality"], progressive=bool(rng.integers(2)), optimize=kw["optimize"], cmyk=True, ycck=bool(rng.integers(2)))
        if it % 2:
            kw["restart_marker_blocks"] = int(rng.integers(1, 40))
    img = content(w, h)
    try:
        data = pil(img, grey, **dict(kw))
    except OSError:                               # Pillow's encoder buffer: some small noisy images at high quality do not fit it
        runs["encoder_refused"] = runs.get("encoder_refused", 0) + 1
        continue
    desc = f"seed {seed} it {it}: {w}x{h} grey={grey} {kw}"
    try:
        want = orc.jpeg_decode(data)
        case("decode", np.array_equal(ctx.jpeg_decode(data), want), desc)
        if not kw.get("cmyk"):
            coef = fennec_amd.Context.jpeg_progressive_coefficients(data)[0]
            case("coefficients", np.array_equal(coef[:, ZIG], orc.jpeg_decode_planes(data, with_coefficients=True)[-1]), desc)
        if it % 5 == 0 and w >= 16 and h >= 16:
            case("recompress", ctx.jpeg_recompress(data, 0.94)[:4] == ctx.jpeg_compress(want, 0.94), desc)
    except Exception as e:                        # noqa: BLE001
        case("decode", False, desc + f" raised {type(e).__name__}: {e}")
    if it % 3 == 0:
        b = bytearray(data)
        lo = data.index(b"\xff\xda")
        for _ in range(int(rng.integers(1, 4))):
            b[int(rng.integers(lo, len(b) - 2))] = int(rng.integers(256))
        bad = bytes(b)
        try:
            w2 = orc.jpeg_decode(bad)
        except Exception:                         # noqa: BLE001
            w2 = None
        try:
            g2 = ctx.jpeg_decode(bad)
        except fennec_amd.FennecError:
            g2 = None
        if g2 is not None and w2 is not None:
            case("damaged_both", np.array_equal(g2, w2), desc + " (damaged)")
        else:
            runs["damaged_refused"] = runs.get("damaged_refused", 0) + 1

print(f"seed {seed}: {it} iterations in {budget:.0f} s, cases {runs}, failures: {len(fails)}")
for f in fails[:20]:
    print("  ", f)
sys.exit(1 if fails else 0)
