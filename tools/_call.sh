set -u
mkdir -p gpurun_out
for rep in 1 2 3; do
echo "== v2"; python tools/time_ssim.py 2>&1 | grep -v amdgpu | head -2
echo "== v1"; FENNEC_HIP_LIB=$PWD/fennec_amd/libfennec_hip_ab.so python tools/time_ssim.py 2>&1 | grep -v amdgpu | head -2
done
