#!/bin/bash
# The five-call evaluation loops behind profiles/r03_eval_loops.json (BASELINE's configs 3, 4, 5, the reference's own
# benchmark size and the headline size) and the headline bench at 4e6 supports, from one tree.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/eval_loops; rm -rf $O; mkdir -p $O
cd $R
i=0
for spec in "--workload farmer --supports 100000" "--workload opf --supports 10000" "--workload quadrotor --supports 16000" "--workload pandemic" "--workload quadrotor --supports 1000000"; do
  i=$((i+1))
  timeout -k 10 400 python3 tools/eval_loop.py $spec > $O/loop_$i.json 2> $O/loop_$i.err || { echo "FAILED $spec"; tail -3 $O/loop_$i.err; exit 1; }
done
timeout -k 10 500 python3 bench.py --supports 4000000 --no-cpu-baseline --no-variants > $O/bench_4e6.json 2> $O/bench_4e6.err || exit 1
echo ok
