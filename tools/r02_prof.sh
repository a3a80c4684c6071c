#!/bin/bash
# round-2 profiles of the final code: (1) kernel-trace/stats of the default bench (tuner on: the second code object's kernels are
# tagged _b48), (2) PMC FETCH_SIZE / WRITE_SIZE in separate passes, (3) the bench line of the same command
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02_prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/stats_bench.json 2> $OUT/stats_bench.err || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --opt autotune=0 > $OUT/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --opt autotune=0 > $OUT/pmc_write.log 2>&1 || exit 1
cd $R
python3 tools/summarize_prof.py $OUT $OUT/summary.json > $OUT/summary.log 2>&1
tail -25 $OUT/summary.log
find $OUT/stats -name "*kernel_stats.csv" | head -2
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1); head -12 $f | cut -c1-160
tail -1 $OUT/stats_bench.json | cut -c1-300
