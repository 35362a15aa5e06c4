#!/bin/bash
# Run ON THE GPU BOX (via gpurun): kernel statistics and PMC counters of any command.
#   tools/pmc.sh TAG "python tools/time_fx.py" [kernel-name filter ...]
# Passes (counters in their own runs, never with --stats: the pool's gpurun refuses some mixes):
#   stats: --kernel-trace --stats        sq1 / sq2: SQ issue + wait + LDS / VMEM counters
#   fetch, write: FETCH_SIZE, WRITE_SIZE (TCC slots do not fit one pass)
# Output: gpurun_out/pmc_$TAG/{stats,sq1,sq2,fetch,write}/ + a printed per-kernel summary.
set -u
TAG=$1; CMD=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o p -- $CMD > "$OUT/stats.out" 2> "$OUT/stats.log"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_SALU --kernel-trace --output-format csv -d "$OUT/sq1" -o p -- $CMD > /dev/null 2> "$OUT/sq1.log"
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d "$OUT/sq2" -o p -- $CMD > /dev/null 2> "$OUT/sq2.log"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -o p -- $CMD > /dev/null 2> "$OUT/fetch.log"
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/write" -o p -- $CMD > /dev/null 2> "$OUT/write.log"
STATS=$(find "$OUT/stats" -name "*kernel_stats.csv" | head -1)
echo "== kernel stats ($STATS)"; head -12 "$STATS" | cut -c1-170
for p in sq1 sq2 fetch write; do echo "== $p"; python tools/pmc_summary.py "$OUT/$p" "$@" | cut -c1-150; done
