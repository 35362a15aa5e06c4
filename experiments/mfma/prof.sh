#!/bin/bash
# on the GPU box: counters of the prototype (own passes)
export TMPDIR=/tmp
cd "$(dirname "$0")"
OUT=../../gpurun_out/mfma_prof; rm -rf $OUT; mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -oE '\b(SQ|TA|TCP|TCC|TD)_[A-Z0-9_]+' | sort -u > $OUT/counters.txt
run() { tag=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$tag -o p -- ./blur_mfma_proto prof > /dev/null 2> $OUT/$tag.log; python3 ../../tools/pmc_summary.py $OUT/$tag blur_mfma | cut -c1-120; }
run sq1 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_SALU
run sq2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT
run sq3 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_I8 SQ_LDS_DATA_FIFO_FULL
run ta TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
