#!/bin/bash
# Everything profiles/<round>_* is made from, in one go on the GPU box:  gpurun -- 'bash tools/collect_round.sh r02'
set -u
R=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
O=gpurun_out
mkdir -p $O
# every collection is summarised here, on the box, into $O/profiles_new/ (copy its files into profiles/ afterwards); the raw
# counter passes are deleted -- together they exceed what gpurun merges back
export FNX_PROFILES_DST=$ROOT/$O/profiles_new
mkdir -p $FNX_PROFILES_DST
collect() {   # tag, bench flags, images per launch
    bash tools/collect_profiles.sh "$1" "$2" > $O/collect_$1.log 2>&1
    python tools/summarise_profiles.py "$1" "$3" > /dev/null 2>> $O/collect_$1.log
    rm -rf $O/profile_$1
}
STEPS=10 collect ${R}_onepass "" 32
collect ${R}_config4 "--workload config4" 1
collect ${R}_config3 "--workload config3 --contexts 1" 4      # (r6: 4 images per batched call, bench.py --config3-chunk)
# the resize kernels config 3's default line does not show: photograph-like content takes resize_mfma_kernel +
# resize_fused_sparse_kernel; one PMC pass set over tools/time_resize.py on both kinds of content
for kind in ramp soft; do
    arg=""; [ $kind = soft ] && arg="soft"
    FNX_TR_CASES=0,1 bash tools/pmc.sh rz_$kind "python tools/time_resize.py $arg" resize > $FNX_PROFILES_DST/${R}_resize_${kind}_counters.txt 2>&1
    rm -rf $O/pmc_rz_$kind
done
python bench.py --workload config3 > $FNX_PROFILES_DST/${R}_config3_contexts4_bench_plain.json 2> $O/c3c4.log
python bench.py --blur-mode exact --no-cpu-baseline --no-batch --no-extras > $FNX_PROFILES_DST/${R}_onepass_exact_bench_plain.json 2>> $O/c3c4.log
python bench.py --pipeline two-call --no-cpu-baseline --no-batch --no-extras > $FNX_PROFILES_DST/${R}_twocall_bench_plain.json 2>> $O/c3c4.log
python bench.py --workload config5 --steps 3 --warmup 1 --no-cpu-baseline > $FNX_PROFILES_DST/${R}_config5_bench_plain.json 2>> $O/c3c4.log
python bench.py --workload config5 --device-search --steps 3 --warmup 1 --no-cpu-baseline > $FNX_PROFILES_DST/${R}_config5_device_search_bench_plain.json 2>> $O/c3c4.log
python bench.py --workload config5 --device-codec --steps 3 --warmup 1 --no-cpu-baseline > $FNX_PROFILES_DST/${R}_config5_device_codec_bench_plain.json 2>> $O/c3c4.log
mkdir -p $O/profile_${R}_config5_device_codec
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$O/profile_${R}_config5_device_codec/stats -o bench -- \
    python $ROOT/bench.py --workload config5 --device-codec --steps 2 --warmup 1 --no-cpu-baseline > $ROOT/$O/profile_${R}_config5_device_codec/bench_under_stats.json 2> $ROOT/$O/c5stats.log )
cp $(find $O/profile_${R}_config5_device_codec -name "*kernel_stats.csv" | head -1) $FNX_PROFILES_DST/${R}_config5_device_codec_kernel_stats.csv; rm -rf $O/profile_${R}_config5_device_codec
python bench.py --workload config5 --device-decode --no-cpu-baseline > $FNX_PROFILES_DST/${R}_config5_device_decode_bench_plain.json 2>> $O/c3c4.log
mkdir -p $O/profile_${R}_config5_device_decode
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$O/profile_${R}_config5_device_decode/stats -o bench -- \
    python $ROOT/bench.py --workload config5 --device-decode --workers 1 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $ROOT/$O/profile_${R}_config5_device_decode/bench_under_stats.json 2> $ROOT/$O/c5dstats.log )
cp $(find $O/profile_${R}_config5_device_decode -name "*kernel_stats.csv" | head -1) $FNX_PROFILES_DST/${R}_config5_device_decode_kernel_stats.csv; rm -rf $O/profile_${R}_config5_device_decode
python tools/time_jpeg_decode.py > $FNX_PROFILES_DST/${R}_time_jpeg_decode.txt 2>&1
python tools/time_batch_jpeg_native.py > $FNX_PROFILES_DST/${R}_time_batch_jpeg_native.txt 2>&1
python tools/time_ops.py > $FNX_PROFILES_DST/${R}_time_ops.txt 2>&1
./tools/time_ops_native > $FNX_PROFILES_DST/${R}_time_ops_native.txt 2>&1
python tools/time_fx.py > $FNX_PROFILES_DST/${R}_time_fx.txt 2>&1
python tools/time_resize.py > $FNX_PROFILES_DST/${R}_time_resize_ramp.txt 2>&1
python tools/time_resize.py soft > $FNX_PROFILES_DST/${R}_time_resize_soft.txt 2>&1
python tools/time_jpeg.py > $FNX_PROFILES_DST/${R}_time_jpeg.txt 2>&1
tail -2 $FNX_PROFILES_DST/${R}_time_jpeg.txt
for t in onepass config4 config3; do tail -c 400 $FNX_PROFILES_DST/${R}_${t}_bench_plain.json; echo; done
du -sh $O
