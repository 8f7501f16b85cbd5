#!/bin/bash
# Run on the GPU box: store-path counters of the encode step's two kernels (which queue holds the alpha0 kernel's stores back?).
# separate --pmc passes with --kernel-trace only.  usage: tools/pmc_store.sh <tag>
set -u
TAG=${1:-r04}
OUT=$PWD/gpurun_out/pmc_store_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python $PWD/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-ksvd --no-aux"
cd /tmp
timeout 180 rocprofv3 --kernel-trace --pmc SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/pmc_sqv -o pmc -- $BENCH > /dev/null 2> $OUT/a.err
# (passes with TA_* or several TCC_*_sum / TCP_*_sum counters abort rocprofv3 on this box (signal 6) and hang until killed: removed;
# the SQ pass above is the one that answers the question)
cd - > /dev/null
LYS_SUMMARY_TOP=4 python $PWD/tools/summarize_profile.py $OUT > $PWD/gpurun_out/pmc_store_${TAG}_summary.txt 2>&1
find $OUT -name "*.db" -delete
cat $PWD/gpurun_out/pmc_store_${TAG}_summary.txt | cut -c1-400
