#!/bin/bash
# usage: isa.sh file.hip [extra flags] -> /tmp/isa/<name>.s
set -e
f=$1; shift
n=$(basename $f .hip)
mkdir -p /tmp/isa
extra=""
case $n in blur_mfma|resize_mfma) extra="-mllvm -amdgpu-mfma-vgpr-form -fno-strict-aliasing";; esac
cd /root/repo/fennec_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden -Wall -Wno-unused-result --offload-arch=gfx950 $extra "$@" --cuda-device-only -S -o /tmp/isa/$n.s $f
