#!/bin/bash
export TMPDIR=/tmp
cd "$(dirname "$0")"
OUT=../../gpurun_out/mfma_prof2; rm -rf $OUT; mkdir -p $OUT
run() { tag=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$tag -o p -- ./blur_mfma_proto prof > /dev/null 2> $OUT/$tag.log; python3 ../../tools/pmc_summary.py $OUT/$tag blur_mfma | cut -c1-120; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum
run tcc2 TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum TCC_REQ_sum TCC_WRITE_sum
run tcp TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
grep -c . $OUT/*.log | head
