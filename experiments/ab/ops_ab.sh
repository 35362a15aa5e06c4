#!/bin/bash
# same-box A/B of per-op timings: experiments/ab/ops_ab.sh libA.so libB.so
cd "$(dirname "$0")/../.."
for lib in "$@"; do
  echo "=== $lib"
  FENNEC_HIP_LIB=$PWD/experiments/ab/$lib python tools/time_resize.py 2>/dev/null
  FENNEC_HIP_LIB=$PWD/experiments/ab/$lib python tools/time_ops.py 2>/dev/null | grep -i "exact\|lanczos\|SSIM\|Analyze\|box"
  FENNEC_HIP_LIB=$PWD/experiments/ab/$lib python bench.py --workload config3 --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | cut -c1-120
done
