set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc2
mkdir -p $OUT
BENCH="python $PWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ksvd"
cd /tmp
rocprofv3 --kernel-trace --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum -d $OUT/a -o pmc -- $BENCH > /dev/null 2> $OUT/a.err
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum -d $OUT/b -o pmc -- $BENCH > /dev/null 2> $OUT/b.err
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_TAG_STALL_sum TCC_REQ_sum TCC_BUSY_sum -d $OUT/c -o pmc -- $BENCH > /dev/null 2> $OUT/c.err
cd - >/dev/null
python tools/summarize_profile.py $OUT 2>&1 | grep -E "PMC pass|bomp_wave|alpha0_n64"
find $OUT -name "*.db" -delete
