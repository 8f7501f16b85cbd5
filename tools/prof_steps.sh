#!/bin/bash
# rocprofv3 kernel trace of tools/ksvd_bench.py + per-launch split of the block-sweep kernels
set -u
OUT=$PWD/gpurun_out/prof_steps
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
rocprofv3 --kernel-trace -d $OUT/trace -o trace -- python $ROOT/tools/ksvd_bench.py 1048576 3 > $OUT/cmd.out 2> $OUT/trace.err
cd $ROOT
tail -1 $OUT/cmd.out
python tools/step_durations.py $OUT "$@"
find $OUT -name "*.db" -delete
