#!/bin/bash
# same-box A/B of library builds with tools/time_blur_kernel.py: experiments/ab/time_ab.sh "args" lib1.so lib2.so ...
cd "$(dirname "$0")/../.."
ARGS=$1; shift
for round in 1 2; do
  for lib in "$@"; do
    echo "$lib: $(FENNEC_HIP_LIB=$PWD/experiments/ab/$lib python tools/time_blur_kernel.py $ARGS | tr '\n' '|')"
  done
done
