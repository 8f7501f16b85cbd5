#!/bin/bash
# like prof_steps.sh but with the number of signals as $1 (fixed per-launch cost of the block sweep at tiny N)
set -u
N=${1:-2048}
OUT=$PWD/gpurun_out/prof_steps_n
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
rocprofv3 --kernel-trace -d $OUT/trace -o trace -- python $ROOT/tools/ksvd_bench.py $N 3 > $OUT/cmd.out 2> $OUT/trace.err
cd $ROOT
tail -1 $OUT/cmd.out
python tools/step_durations.py $OUT 257
find $OUT -name "*.db" -delete
