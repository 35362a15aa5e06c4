#!/usr/bin/env python3
"""fnx_jpeg_decode of ONE 4K file (libjpeg q=90 4:2:0 of synth.large_photo) repeated, result left on the device -- for
rocprofv3 passes over the decoder's kernels (tools/pmc.sh).  python tools/decode_loop.py [reps]"""
import io
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fennec_amd  # noqa: E402
from fennec_amd import synth  # noqa: E402
from PIL import Image  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
b = io.BytesIO()
Image.fromarray(np.ascontiguousarray(synth.large_photo(3840, 2160, 1)[..., :3]), "RGB").save(b, "JPEG", quality=90, subsampling=2)
data = b.getvalue()
ctx = fennec_amd.Context(0)
for _ in range(reps):
    t = ctx.jpeg_decode(data, device=True)
ctx.sync()
