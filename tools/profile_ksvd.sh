#!/bin/bash
# Run on the GPU box: rocprofv3 of one configs[1] alternation loop (tools/ksvd_bench.py): kernel trace / stats, separate
# PMC passes for the HBM traffic of the block-sweep kernel, and the per-launch X/Y split.  -> gpurun_out/prof_ksvd_<tag>*
set -u
TAG=${1:-r02}
OUT=$PWD/gpurun_out/prof_ksvd_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
CMD="python $ROOT/tools/ksvd_bench.py 1048576 3"
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace_cmd.out 2> $OUT/trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $CMD > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_write -o pmc -- $CMD > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $OUT/pmc_sq -o pmc -- $CMD > /dev/null 2> $OUT/pmc_sq.err
cd $ROOT
{
  echo "# command: $CMD  (3 alternations; the sweep = ${STEPS:-129} launches of bksvd_step_kernel per alternation + the index build + the final pass)"
  tail -3 $OUT/trace_cmd.out
  python $ROOT/tools/summarize_profile.py $OUT
  echo
  echo "== per-launch durations of the last sweep: X(0), then the merged launches c = 1 .. nb (STEPS=257 with LYS_BKSVD_MERGED=0: X(c), Y(c)) =="
  python $ROOT/tools/step_durations.py $OUT/trace ${STEPS:-129}
} > $ROOT/gpurun_out/prof_ksvd_${TAG}_summary.txt 2>&1
find $OUT -name "*.db" -delete
tail -40 $ROOT/gpurun_out/prof_ksvd_${TAG}_summary.txt
