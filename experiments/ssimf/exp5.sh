for pf in 8 2; do
make -s -C fennec_amd/csrc DEVELOP=1 BUILD=build_pf$pf OUT=../libfennec_hip_pf$pf.so EXTRA=-DWMF_PF_N=$pf -j16 2>&1 | tail -3
echo "== PF $pf"; FENNEC_HIP_LIB=fennec_amd/libfennec_hip_pf$pf.so python experiments/ssimf/check.py 8k 2>&1 | grep "ramp/adaptive"
done
