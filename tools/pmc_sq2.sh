#!/bin/bash
# Run on the GPU box: second SQ counter pass of the encode step (which pipe of the greedy kernel is hot besides the VALU?)
set -u
OUT=$PWD/gpurun_out/pmc_sq2
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python $PWD/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-ksvd --no-aux"
cd /tmp
timeout 180 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES -d $OUT/pmc_a -o pmc -- $BENCH > /dev/null 2> $OUT/a.err
timeout 180 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $OUT/pmc_b -o pmc -- $BENCH > /dev/null 2> $OUT/b.err
cd - > /dev/null
LYS_SUMMARY_TOP=4 python $PWD/tools/summarize_profile.py $OUT 2>&1 | grep -E "PMC pass|bomp_wave2|alpha0_n64" | cut -c1-500
find $OUT -name "*.db" -delete
