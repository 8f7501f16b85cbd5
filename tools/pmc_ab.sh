#!/bin/bash
# SQ issue/stall counters of the greedy-kernel variants (separate --pmc passes, each bounded by `timeout`).
# usage: tools/pmc_ab.sh <tag> <variant ...>   -> gpurun_out/pmc_<tag>.txt
set -u
TAG=$1; shift
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
CMD="python $ROOT/tools/omp_ab2.py 262144 $*"
cd /tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/a -o pmc -- $CMD > /dev/null 2> $OUT/a.err
timeout 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU -d $OUT/b -o pmc -- $CMD > /dev/null 2> $OUT/b.err
timeout 200 rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_IFETCH SQ_WAVES -d $OUT/c -o pmc -- $CMD > /dev/null 2> $OUT/c.err
cd $ROOT
python tools/summarize_profile.py $OUT 2>&1 | grep -E "PMC pass|bomp_wave" > $ROOT/gpurun_out/pmc_$TAG.txt
find $OUT -name "*.db" -delete
cat $ROOT/gpurun_out/pmc_$TAG.txt
