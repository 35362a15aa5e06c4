set -u
mkdir -p gpurun_out
python tools/fuzz_gpu.py 600 52 > gpurun_out/c15_fuzz_gpu.txt 2>&1; tail -3 gpurun_out/c15_fuzz_gpu.txt
python tools/fuzz_resize.py 120 12 > gpurun_out/c15_fuzz_resize.txt 2>&1; tail -2 gpurun_out/c15_fuzz_resize.txt
python tools/fuzz_resize21.py 120 9 > gpurun_out/c15_fuzz21.txt 2>&1; tail -2 gpurun_out/c15_fuzz21.txt
