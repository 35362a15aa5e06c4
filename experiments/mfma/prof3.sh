#!/bin/bash
# usage: prof3.sh <binary> <arg>
export TMPDIR=/tmp
cd "$(dirname "$0")"
BIN=$1; ARG=$2
OUT=../../gpurun_out/prof_$BIN; rm -rf $OUT; mkdir -p $OUT
run() { tag=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$tag -o p -- ./$BIN $ARG > /dev/null 2> $OUT/$tag.log; }
run sq1 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_SALU
run sq2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT
run sq3 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL
run tcp TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
python3 - <<'PY'
import csv, collections, glob, os
out = os.environ.get('OUTDIR')
PY
python3 - $OUT <<'PY'
import csv, collections, sys
root = sys.argv[1]
for tag in ['sq1','sq2','sq3','tcp']:
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    try:
        rows = list(csv.DictReader(open(f'{root}/{tag}/p_counter_collection.csv')))
    except Exception as e:
        print(tag, 'missing', e); continue
    for r in rows:
        per[int(r['Dispatch_Id'])][r['Counter_Name']] += float(r['Counter_Value'])
    ids = sorted(per)
    # the first timed full-mode dispatches: take dispatch index 2 (after warm-up)
    i = ids[2] if len(ids) > 2 else ids[0]
    print(tag, 'dispatch', i, {k: int(v) for k, v in sorted(per[i].items())})
PY
