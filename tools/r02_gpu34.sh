#!/bin/bash
# det_scatter (plan-driven gather instead of float atomics): GPU suite, determinism probes, product timings with and without
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_s34
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -2 $O/pytest_gpu.log
for a in "quadrotor_oc3 200000" "kinetic 20000"; do timeout -k 10 120 python3 tools/determinism_probe.py $a 2>&1 | tail -1; done
for ds in 1 0; do
  for w in "quadrotor_oc3 500000" "kinetic 200000" "quadrotor 1000000"; do
    set -- $w
    timeout -k 10 250 python3 tools/eval_loop.py --workload $1 --supports $2 --products --opt det_scatter=$ds > $O/$1_ds$ds.json 2>$O/$1_ds$ds.err || echo fail $1 $ds
  done
  timeout -k 10 250 python3 tools/eval_loop.py --workload pandemic --products --opt det_scatter=$ds > $O/pandemic_ds$ds.json 2>$O/pandemic_ds$ds.err || echo fail pandemic
done
python3 - <<PY
import json
for n in ("quadrotor_oc3","kinetic","quadrotor","pandemic"):
    for ds in (0,1):
        j=json.loads(open("$O/%s_ds%d.json"%(n,ds)).read().strip().splitlines()[-1])
        print(n, "det_scatter", ds, "build_s", round(j["build_s"],1), {k:round(j["ms"][k]*1e3,1) for k in ("grad","jprod","jtprod","hprod")})
PY
