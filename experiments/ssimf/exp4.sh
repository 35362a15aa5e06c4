for abl in 1 2 3 4; do
make -s -C fennec_amd/csrc DEVELOP=1 BUILD=build_abl$abl OUT=../libfennec_hip_abl$abl.so EXTRA=-DWM2F_ABL=$abl -j16 2>&1 | tail -3
echo "== ABL $abl"; FENNEC_HIP_LIB=fennec_amd/libfennec_hip_abl$abl.so python experiments/ssimf/check.py 8k 2>&1 | grep "ramp/adaptive"
done
