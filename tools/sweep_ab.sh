#!/bin/bash
# Run on the GPU box: timing of the configs[1] alternation (tools/ksvd_bench.py) + X/Y split of the block-sweep launches
# from one rocprofv3 kernel trace.  usage: tools/sweep_ab.sh <tag>
set -u
TAG=${1:-ab}
OUT=$PWD/gpurun_out/sweep_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
python $ROOT/tools/ksvd_bench.py 1048576 4 2>/dev/null | tail -2
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $ROOT/tools/ksvd_bench.py 1048576 3 > /dev/null 2> $OUT/trace.err
cd $ROOT
python $ROOT/tools/step_durations.py $OUT/trace ${STEPS:-129}
find $OUT -name "*.db" -delete
