set -u
mkdir -p gpurun_out
bash tools/collect_round.sh r05 > gpurun_out/collect_round.log 2>&1
tail -12 gpurun_out/collect_round.log
ls gpurun_out/profiles_new | head -60
