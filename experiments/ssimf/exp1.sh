make -s -C fennec_amd/csrc DEVELOP=1 BUILD=build_dev OUT=../libfennec_hip_dev.so -j16 2>&1 | tail -3
export FENNEC_HIP_LIB=fennec_amd/libfennec_hip_dev.so
for wps in 2 3; do for m2 in 8 12 16 24; do
echo "== WPS $wps M2_WAVES $m2"; FNX_SSIM_F_WAVES=$wps FNX_SSIM_M2_WAVES=$m2 python experiments/ssimf/check.py 8k 2>&1 | grep "ramp/adaptive\|photo/blur2"
done; done
