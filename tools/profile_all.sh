#!/bin/bash
# Run on the GPU box: every profile of a round in one call.  usage: tools/profile_all.sh <tag>  (e.g. r04)
# -> gpurun_out/prof_<tag>_summary.txt, prof_aux_<tag>_summary.txt, prof_ksvd_<tag>_summary.txt, prof_exact_<tag>_summary.txt,
#    kernel_durations_<tag>.json, kernel_durations_aux_<tag>.json, bench_<tag>.json
set -u
TAG=${1:-r04}
mkdir -p gpurun_out
python bench.py > gpurun_out/bench_${TAG}.json 2> gpurun_out/bench_${TAG}.err; echo "bench rc=$?"
bash tools/profile.sh $TAG > /dev/null 2>&1; echo "profile rc=$?"
bash tools/profile_aux.sh $TAG > /dev/null 2>&1; echo "profile_aux rc=$?"
bash tools/profile_ksvd.sh $TAG > /dev/null 2>&1; echo "profile_ksvd rc=$?"
bash tools/prof_cmd.sh exact_$TAG python $PWD/tools/ksvd_bench.py 1048576 3 exact > /dev/null 2>&1; echo "profile_exact rc=$?"
ls -la gpurun_out/*${TAG}* | head -20
