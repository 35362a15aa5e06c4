#!/bin/bash
# tools/fuzz_jpeg_host.cpp under ASan + UBSan over Pillow-written baseline and progressive files (CPU only, ~1 min):
#   bash tools/fuzz_jpeg_host.sh [iters-per-file] > profiles/rNN_fuzz_jpeg_host.txt
set -eu
cd "$(dirname "$0")/.."
T=$(mktemp -d)
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=undefined -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include \
    -Ifennec_amd/csrc tools/fuzz_jpeg_host.cpp fennec_amd/csrc/jpeg_parse.cpp fennec_amd/csrc/jpeg_prog.cpp -o "$T/fuzz"
python - "$T" <<'P'
import sys, numpy as np
from PIL import Image
rng = np.random.default_rng(5)
y, x = np.mgrid[0:72, 0:120]
img = np.clip(np.stack([128 + 60 * np.sin(x / 17.0) * np.cos(y / 11.0) + rng.normal(0, s, x.shape) for s in (6, 9, 12)], -1), 0, 255).astype(np.uint8)
k = 0
for kw in (dict(subsampling=2, progressive=True), dict(subsampling=0, progressive=True, optimize=True), dict(subsampling=1, progressive=True, quality=97),
           dict(subsampling=0, progressive=True, restart_marker_blocks=3), dict(subsampling=2), dict(subsampling=0, restart_marker_blocks=2)):
    Image.fromarray(img).save(f"{sys.argv[1]}/f{k}.jpg", "JPEG", **{"quality": 85, **kw}); k += 1
Image.fromarray(img[..., 0]).save(f"{sys.argv[1]}/f{k}.jpg", "JPEG", quality=85, progressive=True, restart_marker_blocks=4)
k += 1
for kw in (dict(), dict(progressive=True)):
    Image.fromarray(img).convert("CMYK").save(f"{sys.argv[1]}/f{k}.jpg", "JPEG", quality=85, **kw); k += 1
P
ASAN_OPTIONS=detect_leaks=1 "$T/fuzz" "${1:-20000}" "$T"/f*.jpg
rm -rf "$T"
