#!/bin/bash
# A/B of an experimental library build against the shipped one on the headline step (one GPU call):
#   gpurun -- 'bash tools/ab_lib.sh fennec_amd/libfennec_hip_ab.so   (make BUILD=build_ab OUT=../libfennec_hip_ab.so EXTRA=-D...)'
set -u
cd "$(dirname "$0")/.."
ALT=${1:?path of the alternative library}
for round in 1 2 3; do
    echo "== shipped"; python tools/time_onepass.py 3840 2160 32 2>/dev/null | grep -E "one-pass|1p-exact" | tail -4 | sed -n "1p;3p"
    echo "== $ALT"; FENNEC_HIP_LIB=$PWD/$ALT python tools/time_onepass.py 3840 2160 32 2>/dev/null | grep -E "one-pass|1p-exact" | tail -4 | sed -n "1p;3p"
done
exit 0
echo "== parity"
FENNEC_HIP_LIB=$PWD/$ALT python -m pytest tests/test_blur_mfma_gpu.py -m gpu -x -q -k "one_pass or kept" 2>&1 | tail -2
