#!/bin/bash
# rocprofv3 recipe for the headline bench (run on the GPU box through gpurun).
# kernel-trace/stats and each PMC counter group are collected in SEPARATE passes.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-variants --no-live-traffic > $OUT/stats_bench.log 2>&1 || exit 1
pass() {  # name, counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_$name -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-live-traffic > $OUT/pmc_$name.log 2>&1
}
pass fetch FETCH_SIZE || exit 1
pass write WRITE_SIZE || exit 1
pass sq SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES || echo "sq pass failed"
pass sq2 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS || echo "sq2 pass failed"
python3 $R/tools/summarize_prof.py $OUT $OUT/summary.json > $OUT/summary.log 2>&1
tail -40 $OUT/summary.log
