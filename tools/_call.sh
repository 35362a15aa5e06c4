set -u
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -k "resize or config3 or scan or flat" > gpurun_out/c13_tests.log 2>&1; echo "tests rc $?"; tail -2 gpurun_out/c13_tests.log
python tools/time_resize.py soft 2>&1 | grep -v amdgpu
python tools/fuzz_resize.py 90 11 2>&1 | tail -3
./tools/time_ops_native 2>&1 | tail -2
