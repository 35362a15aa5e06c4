set -u
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/c8_tests.log 2>&1; echo "tests rc $?"
tail -4 gpurun_out/c8_tests.log
python bench.py > gpurun_out/c8_bench.json 2> gpurun_out/c8_bench.err; echo "bench rc $?"
tail -c 1500 gpurun_out/c8_bench.err
python tools/fuzz_gpu.py 240 51 > gpurun_out/c8_fuzz_gpu.txt 2>&1; tail -4 gpurun_out/c8_fuzz_gpu.txt
python tools/fuzz_resize21.py 120 7 > gpurun_out/c8_fuzz21.txt 2>&1; tail -3 gpurun_out/c8_fuzz21.txt
