#!/bin/bash
# Run on the GPU box: rocprofv3 --kernel-trace --stats of an arbitrary command, condensed to a text summary.
# Usage: tools/prof_cmd.sh <tag> <command...>   -> gpurun_out/prof_<tag>_summary.txt
set -u
TAG=$1; shift
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- "$@" > $OUT/cmd.out 2> $OUT/trace.err
cd $ROOT
python $ROOT/tools/summarize_profile.py $OUT > $ROOT/gpurun_out/prof_${TAG}_summary.txt 2>&1
find $OUT -name "*.db" -delete
tail -5 $OUT/cmd.out
cat $ROOT/gpurun_out/prof_${TAG}_summary.txt
