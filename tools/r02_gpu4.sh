#!/bin/bash
# round 2, GPU session 4: whole GPU suite, obj variants (kernel time), placement probe with counters
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_s4
mkdir -p $O
cd $R
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -4 $O/pytest_gpu.log
cd /tmp && export TMPDIR=/tmp
for v in "obj_wgs=1024 obj_unroll=1" "obj_wgs=1024 obj_unroll=2" "obj_wgs=512 obj_unroll=2" "obj_wgs=2048 obj_unroll=2" "obj_wgs=256 obj_unroll=2"; do
  tag=$(echo $v | tr ' =' '__')
  set -- $v
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$tag -- python3 $R/tools/eval_loop.py --workload quadrotor --supports 1000000 --iters 50 --opt $1 --opt $2 > $O/st_$tag.log 2>&1 || echo "rocprof failed $tag"
  f=$(find $O/st_$tag -name "*kernel_stats.csv" | head -1)
  echo "$tag: $(grep iem_obj $f)"
  rm -rf $O/st_$tag
done
pmc() {  # label supports iters counters...
  local label=$1 S=$2 it=$3; shift 3
  rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_$label -- python3 $R/tools/placement_counters.py $S $it > $O/pmc_$label.log 2>$O/pmc_$label.err || { echo "pmc $label failed"; tail -3 $O/pmc_$label.err; return; }
  python3 $R/tools/placement_summary.py $O/pmc_$label.log $O/pmc_$label $label > $O/placement_$label.json 2>>$O/pmc_$label.err || echo "summary $label failed"
  rm -rf $O/pmc_$label
}
python3 $R/tools/placement_counters.py 1000000 30 > $O/placement_plain_1e6.log 2>&1; cat $O/placement_plain_1e6.log | cut -c1-120
pmc utcl1_1e6 1000000 5 TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum
pmc eastall_1e6 1000000 5 TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_sum
pmc chan_1e6 1000000 5 TCC_EA0_WRREQ
pmc tag_1e6 1000000 5 TCC_TAG_STALL_sum TCC_BUBBLE_sum TCC_EA0_WRREQ_64B_sum
pmc utcl1_4e6 4000000 5 TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum
pmc eastall_4e6 4000000 5 TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_sum
pmc chan_4e6 4000000 5 TCC_EA0_WRREQ
python3 $R/tools/placement_counters.py 4000000 20 > $O/placement_plain_4e6.log 2>&1; cat $O/placement_plain_4e6.log | cut -c1-120
ls $O
