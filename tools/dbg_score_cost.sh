export TMPDIR=/tmp
# needs a development build: make -C fennec_amd/csrc clean all DEVELOP=1 (release builds do not read FNX_MFMA_DBG)
for d in 0 1 3 7 23; do
  rm -rf gpurun_out/kk; FNX_MFMA_DBG=$d rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kk -o p -- python tools/time_blur_kernel.py > /dev/null 2>&1
  echo "dbg $d: $(grep 'blur_mfma_kernel<true' gpurun_out/kk/p_kernel_stats.csv | cut -d, -f1-4)"
done
