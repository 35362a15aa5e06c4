set -u
mkdir -p gpurun_out
./experiments/dpp/dpp_probe > gpurun_out/c2_dpp.txt 2>&1
FNX_TR_CASES=0 bash tools/pmc.sh rz2_ramp_down "python tools/time_resize.py" resize > gpurun_out/c2_pmc_down.log 2>&1
rm -rf gpurun_out/pmc_rz2_ramp_down
bash tools/pmc.sh ssim8k "python tools/time_ssim.py" windowed > gpurun_out/c2_pmc_ssim.log 2>&1
rm -rf gpurun_out/pmc_ssim8k
bash tools/pmc.sh fx8k "python tools/time_fx.py" fx_ > gpurun_out/c2_pmc_fx.log 2>&1
rm -rf gpurun_out/pmc_fx8k
python bench.py --workload config3 > gpurun_out/c2_config3.json 2> gpurun_out/c2_config3.err
du -sh gpurun_out
