#!/usr/bin/env python3
"""Wall time of the device JPEG decoder (fnx_jpeg_decode into device memory) and of the whole CompressBatch item body on
the device (fnx_jpeg_recompress), next to the host codec's decode of the same file.
python tools/time_jpeg_decode.py [W H]"""
import io
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fennec_amd  # noqa: E402
from fennec_amd import batch, synth  # noqa: E402


def pil(img, **kw):
    from PIL import Image
    b = io.BytesIO()
    Image.fromarray(np.ascontiguousarray(img[..., :3]), "RGB").save(b, "JPEG", **kw)
    return b.getvalue()


sizes = [(int(sys.argv[1]), int(sys.argv[2]))] if len(sys.argv) > 2 else [(1920, 1080), (3840, 2160)]
ctx = fennec_amd.Context(0)
for W, H in sizes:
    photo = synth.large_photo(W, H, 1)
    for label, data in (("libjpeg q=90 4:2:0", pil(photo, quality=90, subsampling=2)),
                        ("libjpeg q=75 4:2:0", pil(photo, quality=75, subsampling=2)),
                        ("libjpeg q=95 4:4:4", pil(photo, quality=95, subsampling=0)),
                        ("libjpeg q=90 4:2:0 progressive", pil(photo, quality=90, subsampling=2, progressive=True)),
                        ("libjpeg q=95 4:4:4 progressive", pil(photo, quality=95, subsampling=0, progressive=True)),
                        ("bench source (synth.make_test_image, q=90)", pil(synth.make_test_image(W, H), quality=90, subsampling=2))):
        t = ctx.jpeg_decode(data, device=True)
        ctx.sync()
        n = 20
        t0 = time.perf_counter()
        for _ in range(n):
            t = ctx.jpeg_decode(data, device=True)
        ctx.sync()
        dev = (time.perf_counter() - t0) / n
        t0 = time.perf_counter()
        for _ in range(3):
            batch.pillow_decode(data)
        host = (time.perf_counter() - t0) / 3
        r = ctx.jpeg_recompress(data, 0.94)
        t0 = time.perf_counter()
        for _ in range(n):
            r = ctx.jpeg_recompress(data, 0.94)
        rec = (time.perf_counter() - t0) / n
        print(f"{W}x{H} {label:46s} {len(data) / 1e6:6.2f} MB: device decode {dev * 1e3:7.3f} ms ({W * H / dev / 1e6:7.0f} MP/s, "
              f"{len(data) / dev / 1e9:5.2f} GB/s of file), host decode {host * 1e3:7.2f} ms; recompress {rec * 1e3:7.3f} ms -> "
              f"{len(r[0])} bytes q={r[1]} steps={r[3]}", flush=True)
