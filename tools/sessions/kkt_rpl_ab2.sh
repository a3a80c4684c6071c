#!/bin/bash
# second pass of the rows-per-lane A/B (after the in-place restart of the pivot row): 40 x 40 with two rows per lane
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04_rpl2b}; mkdir -p $O; : > $O/table.txt
cd $R
export IEM_KKT_EXPERIMENTS=1
run() {
  timeout -k 10 300 python3 tools/kkt_chain_bench.py --workload $2 --supports $3 --iters 5 --cabi 0 > $O/run.json 2> $O/run.err || { echo "$1 $2: FAILED $(tail -1 $O/run.err)" | tee -a $O/table.txt; return; }
  python3 -c "
import json; j=json.load(open('$O/run.json')); print('$1 $2 $3: factor %.3f ms  solve %.3f  resid %.1e  inertia %s' % (j['ms']['factor'], j['ms']['solve_no_refinement'], j['abs_residual_without'], j['inertia']))" | tee -a $O/table.txt
}
unset IEM_KKT_ROWWISE IEM_KKT_DEFS
run "default" hovercraft 100000
run "default" quadrotor 100000
export IEM_KKT_ROWWISE=2
run "rpl=2,waves=1" quadrotor 100000
export IEM_KKT_DEFS="#define KKT_ROW_WAVES 2"
run "rpl=2,waves=2" quadrotor 100000
run "rpl=2,waves=2" kinetic 100000
run "rpl=2,waves=2" hovercraft 100000
export IEM_KKT_DEFS="#define KKT_ROW_WAVES 3"
run "rpl=2,waves=3" hovercraft 100000
export IEM_KKT_ROWWISE=1 IEM_KKT_DEFS="#define KKT_ROW_WAVES 2"
run "rpl=1,waves=2" quadrotor 100000
