#!/bin/bash
# kkt_eliminate: lane = row form (KKT_ROWWISE) against the matrix-core panel form — parity tests, then timings at scale
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04_rowwise}; mkdir -p $O
cd $R
timeout -k 10 500 python -m pytest tests/test_kkt_chain.py tests/test_kkt_cabi.py -x -q -m gpu > $O/tests_default.log 2>&1 || { tail -20 $O/tests_default.log; exit 1; }
tail -2 $O/tests_default.log
export IEM_KKT_EXPERIMENTS=1
IEM_KKT_ROWWISE=1 timeout -k 10 500 python -m pytest tests/test_kkt_chain.py -x -q -m gpu -k "test_chain_kkt_on_gpu" > $O/tests_rowwise_all.log 2>&1 || { tail -20 $O/tests_rowwise_all.log; exit 1; }
tail -2 $O/tests_rowwise_all.log
for wl in "quadrotor 100000" "hovercraft 100000" "kinetic 100000" "quadrotor 1000000"; do
  set -- $wl
  for rw in 0 1; do
    IEM_KKT_ROWWISE=$rw timeout -k 10 300 python3 tools/kkt_chain_bench.py --workload $1 --supports $2 --iters 5 > $O/run.json 2> $O/run.err || { tail -3 $O/run.err; exit 1; }
    python3 - "$1 $2 rowwise=$rw" $O/run.json <<'PY' | tee -a $O/table.txt
import json, sys
j = json.load(open(sys.argv[2]))
print(sys.argv[1], json.dumps({k: v for k, v in j.items() if k in ("ms", "cabi_ms", "abs_residual", "inertia", "layout")}))
PY
  done
done
unset IEM_KKT_ROWWISE IEM_KKT_EXPERIMENTS
timeout -k 10 200 python3 tools/kkt_hub_bench.py --iters 3 > $O/hub.json 2> $O/hub.err || { tail -3 $O/hub.err; exit 1; }
cat $O/hub.json
