#!/bin/bash
# axis sums (det_axis): GPU suite, pandemic products with and without
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_s32
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -2 $O/pytest_gpu.log
for da in 1 0; do
  timeout -k 10 200 python3 tools/eval_loop.py --workload pandemic --products --opt det_axis=$da > $O/pandemic_da$da.json 2>$O/p$da.err || echo fail p
  python3 -c "
import json; j=json.loads(open('$O/pandemic_da$da.json').read().strip().splitlines()[-1]); print('det_axis', $da, {k:round(j['ms'][k]*1e3,1) for k in j['ms']})"
done
