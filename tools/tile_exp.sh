mkdir -p gpurun_out/r4b
B="python bench.py --steps 60 --warmup 5 --no-aux --no-ksvd --no-cpu-baseline"
run() { tag=$1; shift; env "$@" $B 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = j['roofline']
print('$tag', 'ms/step %.3f  value %.1fM  greedy %.3f ms  gemm %.3f ms  whole %.4f' % (j['ms_per_step'], j['value']/1e6, r['avg_launch_ms'], r['gemm_stage']['avg_launch_ms'], r['whole_step']['frac']))
"; }
run default A=1
run tile1024 LYS_TILE_MB=1024
run tile512 LYS_TILE_MB=512
run tile256 LYS_TILE_MB=256
run tile128 LYS_TILE_MB=128
run tile64 LYS_TILE_MB=64
run pipe_tile1024 LYS_TILE_MB=1024 LYS_PIPELINE=1
run pipe_tile256 LYS_TILE_MB=256 LYS_PIPELINE=1
run pipe_tile128 LYS_TILE_MB=128 LYS_PIPELINE=1
run pipe_tile64 LYS_TILE_MB=64 LYS_PIPELINE=1
run default2 A=1
