#!/bin/bash
# A/B of environment-selected variants through short bench.py runs inside ONE job (same box): tools/ab_env.sh VAR v1 v2 [reps]
VAR=$1; A=$2; B=$3; REPS=${4:-2}
BENCH="python bench.py --steps 60 --warmup 5 --no-aux --no-ksvd --no-cpu-baseline"
run() { env $VAR=$1 $BENCH 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = j['roofline']
print('$VAR=$1', 'ms/step %.3f  value %.1fM  greedy %.3f ms  gemm %.3f ms  whole %.4f' % (j['ms_per_step'], j['value']/1e6, r['avg_launch_ms'], r['gemm_stage']['avg_launch_ms'], r['whole_step']['frac']))
"; }
for rep in $(seq $REPS); do run $A; run $B; done
