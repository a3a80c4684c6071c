#!/bin/bash
# chain KKT solver: the measurement set behind profiles/r03_kkt_chain.json (bench per workload, Newton demo, per-level trace)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/kkt_suite; rm -rf $O; mkdir -p $O
cd $R
for spec in "farmer 100000" "opf 20000" "quadrotor 100000" "hovercraft 100000" "quadrotor 1000000" "quadrotor_oc3 50000" "kinetic 100000" "pandemic3 100000" "pandemic5 50000"; do
  set -- $spec
  timeout -k 10 400 python3 tools/kkt_chain_bench.py --workload $1 --supports $2 > $O/bench_$1_$2.log 2>&1 || { echo "FAILED $spec"; tail -5 $O/bench_$1_$2.log; exit 1; }
  grep -h '^{' $O/bench_$1_$2.log | cut -c1-700
done
timeout -k 10 400 python3 tools/newton_kkt_demo.py --supports 100000 --iters 25 > $O/newton.log 2>&1 || { tail -5 $O/newton.log; exit 1; }
tail -2 $O/newton.log | cut -c1-600
bash tools/sessions/kkt_prof.sh > $O/prof.log 2>&1; tail -40 $O/prof.log | cut -c1-160
