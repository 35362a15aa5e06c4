#!/usr/bin/env python3
"""The device JPEG decoder against the oracle's on LARGE files (many workgroups, cross-workgroup rounds): random size up to
4K, quality, subsampling / grey, restart intervals, optimised tables; each file intact and with 1-3 damaged scan bytes.
python tools/fuzz_jpeg_big.py [seconds] [seed]"""
import io
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fennec_amd  # noqa: E402
from fennec_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from PIL import Image  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ctx = fennec_amd.Context(0)
runs = {"intact": 0, "damaged_both": 0, "damaged_neither": 0, "damaged_one_sided": 0}
fails = []
t_end = time.time() + budget
it = 0
while time.time() < t_end:
    it += 1
    w, h = int(rng.integers(900, 3841)), int(rng.integers(600, 2161))
    kind = int(rng.integers(3))
    img = synth.large_photo(w, h, int(rng.integers(100))) if kind == 0 else (synth.make_test_image(w, h) if kind == 1 else
                                                                              synth.noise_image(w, h, int(rng.integers(1 << 30))))
    q = int(rng.integers(5, 99))
    sub = int(rng.integers(0, 4))
    rs = {} if rng.integers(2) else ({"restart_marker_rows": int(rng.integers(1, 5))} if rng.integers(2) else
                                     {"restart_marker_blocks": int(rng.integers(1, 300))})
    buf = io.BytesIO()
    try:
        src = Image.fromarray(np.ascontiguousarray(img[..., :3]), "RGB")
        if sub == 3:
            src.convert("L").save(buf, "JPEG", quality=q, optimize=bool(rng.integers(2)) and q < 90, **rs)
        else:
            src.save(buf, "JPEG", quality=q, subsampling=sub, optimize=bool(rng.integers(2)) and q < 90, **rs)
    except OSError:
        continue
    data = buf.getvalue()
    desc = f"seed={seed} it={it} {w}x{h} q={q} sub={sub} {rs} {len(data)} bytes"
    runs["intact"] += 1
    if not np.array_equal(ctx.jpeg_decode(data), orc.jpeg_decode(data)):
        fails.append(("intact", desc))
        print("FAIL intact", desc, flush=True)
    scan = data.index(b"\xff\xda") + (14 if sub != 3 else 10)
    bad = bytearray(data)
    for _ in range(int(rng.integers(1, 4))):
        bad[int(rng.integers(scan, len(bad) - 2))] = int(rng.integers(0, 256))
    bad = bytes(bad)
    try:
        want = orc.jpeg_decode(bad)
    except Exception:
        want = None
    try:
        got = ctx.jpeg_decode(bad)
    except fennec_amd.FennecError:
        got = None
    if got is not None and want is not None:
        runs["damaged_both"] += 1
        if not np.array_equal(got, want):
            fails.append(("damaged_both", desc))
            print("FAIL damaged_both", desc, flush=True)
            open(os.path.join("gpurun_out", f"fuzzbig_fail_{seed}_{it}.jpg"), "wb").write(bad)
    elif (got is None) != (want is None):
        runs["damaged_one_sided"] += 1
    else:
        runs["damaged_neither"] += 1
print("runs:", runs)
print("FAILURES:", len(fails))
for f in fails[:20]:
    print("  ", f)
sys.exit(1 if fails else 0)
