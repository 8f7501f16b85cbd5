#!/bin/bash
# SQ issue/stall counters of the encode step (separate --pmc passes, each bounded by `timeout`).
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc3
rm -rf $OUT; mkdir -p $OUT
BENCH="python $PWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ksvd"
cd /tmp
timeout 150 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/a -o pmc -- $BENCH > /dev/null 2> $OUT/a.err
timeout 150 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU -d $OUT/b -o pmc -- $BENCH > /dev/null 2> $OUT/b.err
timeout 150 rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $OUT/c -o pmc -- $BENCH > /dev/null 2> $OUT/c.err
cd - >/dev/null
python tools/summarize_profile.py $OUT 2>&1 | grep -E "PMC pass|bomp_wave|alpha0_n64"
find $OUT -name "*.db" -delete
