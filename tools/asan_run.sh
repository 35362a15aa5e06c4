#!/bin/bash
# Part of the GPU suite against the AddressSanitizer build of the host layer (make -C fennec_amd/csrc asan):
# the C ABI's argument handling, scratch / plan / table caches, result FIFO and the two-stream bookkeeping run under
# ASan (g++-built host files, gcc's runtime); device code is not instrumented.  Run on the GPU box (the swap below changes the box's scratch copy only):
#   gpurun -- 'bash tools/asan_run.sh'
set -u
cd "$(dirname "$0")/.."
RT=$(gcc -print-file-name=libasan.so)
[ -f fennec_amd/libfennec_hip_asan.so ] || { echo "build it first: make -C fennec_amd/csrc asan"; exit 2; }
cp fennec_amd/libfennec_hip.so /tmp/libfennec_hip.keep
cp fennec_amd/libfennec_hip_asan.so fennec_amd/libfennec_hip.so
# python itself is not instrumented: preload the runtime; leaks are python's and the HIP runtime's own, not ours to chase here
mkdir -p gpurun_out
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1:protect_shadow_gap=0:verify_asan_link_order=0:log_path=gpurun_out/asan_report
# ASan's dlopen interceptor loses the caller's RUNPATH: torch finds its own libraries through LD_LIBRARY_PATH instead
export LD_LIBRARY_PATH=$(python -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), 'lib'))"):${LD_LIBRARY_PATH:-}
# libstdc++ beside the runtime: its __cxa_throw interceptor must find the real one (torch throws during CUDA init)
LD_PRELOAD="$RT $(gcc -print-file-name=libstdc++.so)" timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_jpeg_roundtrip.py tests/test_jpeg_decode.py tests/test_jpeg_progressive.py tests/test_blur_mfma_gpu.py -x -q -m gpu \
    -k "not 8k and not 4k and not config" "$@" > gpurun_out/asan_pytest.log 2>&1
rc=$?
tail -15 gpurun_out/asan_pytest.log; echo "pytest rc=$rc"; ls gpurun_out/asan_report* 2>/dev/null && head -60 gpurun_out/asan_report*
cp /tmp/libfennec_hip.keep fennec_amd/libfennec_hip.so
exit $rc
