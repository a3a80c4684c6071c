#!/bin/bash
# deferred small slots (a few items against many): GPU suite, option matrix, product timings
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_s36
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -2 $O/pytest_gpu.log
timeout -k 10 600 python3 tools/option_matrix_gpu.py > $O/option_matrix.log 2>$O/option_matrix.err; echo "matrix rc=$?"; tail -1 $O/option_matrix.log
for ds in 1 0; do
  timeout -k 10 250 python3 tools/eval_loop.py --workload pandemic --products --opt det_scatter=$ds > $O/pandemic_ds$ds.json 2>$O/p$ds.err || echo fail
  timeout -k 10 250 python3 tools/eval_loop.py --workload hovercraft --supports 1000000 --products --opt det_scatter=$ds > $O/hovercraft_ds$ds.json 2>$O/h$ds.err || echo fail
done
python3 - <<PY
import json
for n in ("pandemic","hovercraft"):
    for ds in (0,1):
        j=json.loads(open("$O/%s_ds%d.json"%(n,ds)).read().strip().splitlines()[-1])
        print(n, "det_scatter", ds, {k:round(j["ms"][k]*1e3,1) for k in ("grad","jprod","jtprod","hprod")})
PY
