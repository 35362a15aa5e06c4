make -s -C fennec_amd/csrc DEVELOP=1 BUILD=build_dev OUT=../libfennec_hip_dev.so -j16 2>&1 | tail -3
export FENNEC_HIP_LIB=fennec_amd/libfennec_hip_dev.so
for wps in 2 3; do
echo "== WPS $wps"; FNX_SSIM_F_WAVES=$wps python experiments/ssimf/check.py 8k 2>&1 | grep "ramp/adaptive\|photo/blur2\|flat250"
done
