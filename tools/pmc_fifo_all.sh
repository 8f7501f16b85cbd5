#!/bin/bash
# Run on the GPU box: SQ -> TA FIFO-full counters of EVERY kernel of the bench (auxiliary legs included): which kernels are
# bound by the number of vector-memory instructions the texture addresser takes?
set -u
OUT=$PWD/gpurun_out/pmc_fifo_all
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python $PWD/bench.py --steps 5 --warmup 1 --no-cpu-baseline"
cd /tmp
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD -d $OUT/pmc_a -o pmc -- $BENCH > /dev/null 2> $OUT/a.err
cd - > /dev/null
LYS_SUMMARY_TOP=60 python $PWD/tools/summarize_profile.py $OUT 2>&1 | cut -c1-330 > $PWD/gpurun_out/pmc_fifo_all_summary.txt
find $OUT -name "*.db" -delete
cat $PWD/gpurun_out/pmc_fifo_all_summary.txt
