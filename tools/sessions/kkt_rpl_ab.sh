#!/bin/bash
# lane-per-row eliminate: rows per lane (KKT_ROWWISE = 1 / 2) and the register budget (KKT_ROW_WAVES), per workload
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04_rpl}; mkdir -p $O; : > $O/table.txt
cd $R
export IEM_KKT_EXPERIMENTS=1
run() {  # label, workload, supports
  timeout -k 10 300 python3 tools/kkt_chain_bench.py --workload $2 --supports $3 --iters 5 --cabi 0 > $O/run.json 2> $O/run.err || { echo "$1 $2: FAILED $(tail -1 $O/run.err)" | tee -a $O/table.txt; return; }
  python3 -c "
import json; j=json.load(open('$O/run.json')); print('$1 $2 $3: factor %.3f ms  solve %.3f  resid %.1e  inertia %s' % (j['ms']['factor'], j['ms']['solve_no_refinement'], j['abs_residual_without'], j['inertia']))" | tee -a $O/table.txt
}
for cfg in "rpl=2" "rpl=2,waves=2" "rpl=1" "rpl=0"; do
  unset IEM_KKT_ROWWISE IEM_KKT_DEFS
  case $cfg in
    "rpl=2") export IEM_KKT_ROWWISE=2 ;;
    "rpl=2,waves=2") export IEM_KKT_ROWWISE=2 IEM_KKT_DEFS="#define KKT_ROW_WAVES 2" ;;
    "rpl=1") export IEM_KKT_ROWWISE=1 ;;
    "rpl=0") export IEM_KKT_ROWWISE=0 ;;
  esac
  run "$cfg" hovercraft 100000
  run "$cfg" quadrotor 100000
  run "$cfg" kinetic 100000
done
