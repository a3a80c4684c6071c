#!/bin/bash
# rocprofv3 kernel stats of the chain KKT bench (program directly after --):  kkt_prof.sh [workload] [supports]
W=${1:-quadrotor}; N=${2:-100000}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_kktprof; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/kkt_chain_bench.py --workload $W --supports $N --iters 5 --cabi 0 > $O/bench.log 2>&1 || exit 1
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); head -12 $f | cut -c1-200
t=$(find $O/stats -name "*kernel_trace.csv" | head -1)
python3 - "$t" <<'PY'
import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"] in ("kkt_eliminate","kkt_update")]
for r in rows[-35:]:
    print(r["Kernel_Name"], int(r["Grid_Size_X"])//int(r["Workgroup_Size_X"]), (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, "us", "vgpr", r["VGPR_Count"], "lds", r["LDS_Block_Size"])
PY
