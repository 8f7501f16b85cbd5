#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel-trace/stats + HBM-byte PMC passes of bench.py WITH its auxiliary legs
# (configs[2] shard through bomp_block_kernel, configs[3] mini-batch through the LARS coder + online-DL update, the K-SVD
# alternation).  Usage: tools/profile_aux.sh <tag>   -> gpurun_out/prof_aux_<tag>_summary.txt
set -u
TAG=${1:-r03}
OUT=$PWD/gpurun_out/prof_aux_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python $PWD/bench.py --steps 5 --warmup 1 --no-cpu-baseline"
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/trace_bench.json 2> $OUT/trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $BENCH > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_write -o pmc -- $BENCH > /dev/null 2> $OUT/pmc_write.err
cd - > /dev/null
{
  echo "rocprofv3 passes of: python bench.py --steps 5 --warmup 1 --no-cpu-baseline   (auxiliary legs included)"
  echo "bench line of the kernel-trace pass:"
  tail -1 $OUT/trace_bench.json
  echo
  LYS_SUMMARY_TOP=40 python $PWD/tools/summarize_profile.py $OUT $PWD/gpurun_out/kernel_durations_aux_${TAG}.json
} > $PWD/gpurun_out/prof_aux_${TAG}_summary.txt 2>&1
find $OUT -name "*.db" -delete
tail -70 $PWD/gpurun_out/prof_aux_${TAG}_summary.txt | cut -c1-220
