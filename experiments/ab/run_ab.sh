#!/bin/bash
# same-box A/B of library builds: experiments/ab/run_ab.sh lib1.so lib2.so ...  (3 interleaved rounds, config-2 default line)
cd "$(dirname "$0")/../.."
for round in 1 2 3; do
  for lib in "$@"; do
    v=$(FENNEC_HIP_LIB=$PWD/experiments/ab/$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-batch $AB_FLAGS | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['avg_launch_ms'])")
    echo "$lib $v"
  done
done
