set -u
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -k "analyze or scan or flat or profile_hook" > gpurun_out/c7_tests.log 2>&1; echo "tests rc $?"
tail -3 gpurun_out/c7_tests.log
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/probe -o p -- python tools/_probe.py > gpurun_out/c4_probe.txt 2>&1
cp $(find gpurun_out/probe -name "*kernel_stats.csv" | head -1) gpurun_out/c4_kernel_stats.csv
rm -rf gpurun_out/probe
cut -c1-140 gpurun_out/c4_kernel_stats.csv
./tools/time_ops_native > gpurun_out/c7_time_ops_native.txt 2>&1
tail -3 gpurun_out/c7_time_ops_native.txt
