#!/bin/bash
# A/B of an experimental library build against the shipped one on the headline step (one GPU call):
#   gpurun -- 'bash tools/ab_presum.sh fennec_amd/libfennec_hip_presum.so'
set -u
cd "$(dirname "$0")/.."
ALT=${1:?path of the alternative library}
for round in 1 2 3; do
    echo "== shipped"; python tools/time_onepass.py 3840 2160 32 2>/dev/null | grep -E "one-pass|1p-exact" | tail -2
    echo "== $ALT"; FENNEC_HIP_LIB=$PWD/$ALT python tools/time_onepass.py 3840 2160 32 2>/dev/null | grep -E "one-pass|1p-exact" | tail -2
done
echo "== parity of the alternative build"
FENNEC_HIP_LIB=$PWD/$ALT python -m pytest tests/test_blur_mfma_gpu.py -m gpu -x -q -k "one_pass or kept" 2>&1 | tail -2
