set -u
mkdir -p gpurun_out
bash tools/asan_run.sh > gpurun_out/asan_run.log 2>&1; echo "asan rc $?"; tail -6 gpurun_out/asan_run.log
bash tools/tsan_run.sh > gpurun_out/tsan_run.log 2>&1; echo "tsan rc $?"; tail -12 gpurun_out/tsan_run.log
ls gpurun_out | head -30
