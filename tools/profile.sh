#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel-trace/stats + separate PMC passes of bench.py.
# Usage: tools/profile.sh <tag>     -> writes gpurun_out/prof_<tag>/... and gpurun_out/prof_<tag>_summary.txt
set -u
TAG=${1:-r01}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python $PWD/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-ksvd --no-aux"
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/trace_bench.json 2> $OUT/trace.err
# PMC passes: counters in their own runs (never combined with sys/hip trace domains)
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $OUT/pmc_sq -o pmc -- $BENCH > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $BENCH > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_write -o pmc -- $BENCH > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $OUT/pmc_mfma -o pmc -- $BENCH > /dev/null 2> $OUT/pmc_mfma.err
cd - > /dev/null
python $PWD/tools/summarize_profile.py $OUT $PWD/gpurun_out/kernel_durations_${TAG}.json > $PWD/gpurun_out/prof_${TAG}_summary.txt 2>&1
find $OUT -name "*.db" -delete   # summaries only: keep the merge under the gpurun_out size cap
tail -60 $PWD/gpurun_out/prof_${TAG}_summary.txt
