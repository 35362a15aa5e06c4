#!/bin/bash
# Everything profiles/<round>_* is made from, in one go on the GPU box:  gpurun -- 'bash tools/collect_round.sh r02'
set -u
R=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
O=gpurun_out
mkdir -p $O
STEPS=10 bash tools/collect_profiles.sh ${R}_onepass "" > $O/collect_onepass.log 2>&1
bash tools/collect_profiles.sh ${R}_config4 "--workload config4" > $O/collect_config4.log 2>&1
bash tools/collect_profiles.sh ${R}_config3 "--workload config3 --contexts 1" > $O/collect_config3.log 2>&1
python bench.py --workload config3 > $O/${R}_config3_contexts4_bench_plain.json 2> $O/c3c4.log
python bench.py --blur-mode exact --no-cpu-baseline --no-batch --no-extras > $O/${R}_onepass_exact_bench_plain.json 2>> $O/c3c4.log
python bench.py --pipeline two-call --no-cpu-baseline --no-batch --no-extras > $O/${R}_twocall_bench_plain.json 2>> $O/c3c4.log
python bench.py --workload config5 --steps 3 --warmup 1 --no-cpu-baseline > $O/${R}_config5_bench_plain.json 2>> $O/c3c4.log
python bench.py --workload config5 --device-search --steps 3 --warmup 1 --no-cpu-baseline > $O/${R}_config5_device_search_bench_plain.json 2>> $O/c3c4.log
python bench.py --workload config5 --device-codec --steps 3 --warmup 1 --no-cpu-baseline > $O/${R}_config5_device_codec_bench_plain.json 2>> $O/c3c4.log
mkdir -p $O/profile_${R}_config5_device_codec
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$O/profile_${R}_config5_device_codec/stats -o bench -- \
    python $ROOT/bench.py --workload config5 --device-codec --steps 2 --warmup 1 --no-cpu-baseline > $ROOT/$O/profile_${R}_config5_device_codec/bench_under_stats.json 2> $ROOT/$O/c5stats.log )
python bench.py --workload config5 --device-decode --no-cpu-baseline > $O/${R}_config5_device_decode_bench_plain.json 2>> $O/c3c4.log
mkdir -p $O/profile_${R}_config5_device_decode
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$O/profile_${R}_config5_device_decode/stats -o bench -- \
    python $ROOT/bench.py --workload config5 --device-decode --workers 1 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $ROOT/$O/profile_${R}_config5_device_decode/bench_under_stats.json 2> $ROOT/$O/c5dstats.log )
python tools/time_jpeg_decode.py > $O/${R}_time_jpeg_decode.txt 2>&1
python tools/time_batch_jpeg_native.py > $O/${R}_time_batch_jpeg_native.txt 2>&1
python tools/time_ops.py > $O/${R}_time_ops.txt 2>&1
python tools/time_fx.py > $O/${R}_time_fx.txt 2>&1
python tools/time_resize.py > $O/${R}_time_resize_ramp.txt 2>&1
python tools/time_resize.py soft > $O/${R}_time_resize_soft.txt 2>&1
python tools/time_jpeg.py > $O/${R}_time_jpeg.txt 2>&1
tail -2 $O/${R}_time_jpeg.txt
for t in onepass config4 config3; do tail -c 400 $O/profile_${R}_$t/bench_plain.json; echo; done
