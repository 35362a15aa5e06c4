import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import fennec_amd
from fennec_amd import synth
import io
from PIL import Image
ctx = fennec_amd.Context(0)
img = synth.large_photo(3840, 2160, 1)
buf = io.BytesIO(); Image.fromarray(np.ascontiguousarray(img[..., :3]), "RGB").save(buf, "JPEG", quality=90, subsampling=2)
data = buf.getvalue()
for _ in range(5): ctx.jpeg_decode(data, to_host=False) if "to_host" in ctx.jpeg_decode.__code__.co_varnames else ctx.jpeg_decode(data)
t0 = time.perf_counter()
N = 30
for _ in range(N): ctx.jpeg_decode(data)
print(len(data), (time.perf_counter() - t0) / N * 1e3, "ms per decode (incl. D2H of the image)")
