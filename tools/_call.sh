set -u
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -k "resize or config3 or msssim or analyze or scan or flat or profile_hook" > gpurun_out/c5_tests.log 2>&1; echo "tests rc $?"
tail -4 gpurun_out/c5_tests.log
python tools/time_resize.py > gpurun_out/c5_time_resize_ramp.txt 2>&1
cat gpurun_out/c5_time_resize_ramp.txt
python tools/time_resize.py soft > gpurun_out/c5_time_resize_soft.txt 2>&1
cat gpurun_out/c5_time_resize_soft.txt
./tools/time_ops_native > gpurun_out/c5_time_ops_native.txt 2>&1
tail -3 gpurun_out/c5_time_ops_native.txt
python bench.py --workload config3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config3:', d['value'], d['ms_per_step'])"
python tools/fuzz_resize21.py 90 5 > gpurun_out/c5_fuzz21.txt 2>&1; tail -5 gpurun_out/c5_fuzz21.txt
