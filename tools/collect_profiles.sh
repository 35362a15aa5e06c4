#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 evidence for bench.py's numbers.
#   1. --kernel-trace --stats of the bench command  -> per-kernel average durations
#   2. separate --pmc passes (FETCH_SIZE, WRITE_SIZE: TCC slots do not fit one pass)
# Everything lands in gpurun_out/profile_$TAG/; tools/summarise_profiles.py turns it into the
# small files committed under profiles/.
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/profile_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
EXTRA=${2:-}      # e.g. "--pipeline one-pass" (use a tag of its own: r01_onepass)
CMD="python $ROOT/bench.py --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline --no-extras --no-batch $EXTRA"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- $CMD > "$OUT/bench_under_stats.json" 2> "$OUT/stats.log"
PCMD="$CMD --prewarm 0"     # counters do not depend on the clock ramp; keep the PMC passes short
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o bench -- $PCMD > /dev/null 2> "$OUT/pmc_fetch.log"
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o bench -- $PCMD > /dev/null 2> "$OUT/pmc_write.log"
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d "$OUT/pmc_sq" -o bench -- $PCMD > /dev/null 2> "$OUT/pmc_sq.log"
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$OUT/pmc_sq2" -o bench -- $PCMD > /dev/null 2> "$OUT/pmc_sq2.log"
python "$ROOT/bench.py" --steps ${PLAIN_STEPS:-20} --warmup 3 $EXTRA > "$OUT/bench_plain.json" 2> "$OUT/bench_plain.log"
find "$OUT" -name "*.csv" | head -20
tail -1 "$OUT/bench_plain.json"
