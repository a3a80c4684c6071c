#!/bin/bash
# pulled stencil neighbours in the scatter kinds: GPU suite, product timings before/after
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_s28
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -2 $O/pytest_gpu.log
for ps in 1 0; do
  timeout -k 10 200 python3 tools/eval_loop.py --workload quadrotor --supports 1000000 --products --opt pull_scatter=$ps > $O/quadrotor_1e6_pull$ps.json 2>$O/q$ps.err || echo fail q
  timeout -k 10 200 python3 tools/eval_loop.py --workload pandemic --products --opt pull_scatter=$ps > $O/pandemic_5e5_pull$ps.json 2>$O/p$ps.err || echo fail p
  timeout -k 10 200 python3 tools/eval_loop.py --workload hovercraft --supports 1000000 --products --opt pull_scatter=$ps > $O/hovercraft_1e6_pull$ps.json 2>$O/h$ps.err || echo fail h
done
python3 - <<PY
import json
for n in ("quadrotor_1e6","pandemic_5e5","hovercraft_1e6"):
    for ps in (0,1):
        j=json.loads(open("$O/%s_pull%d.json"%(n,ps)).read().strip().splitlines()[-1])
        print(n, "pull", ps, {k:(round(j["ms"][k]*1e3,1), round(j["alg_bytes"][k]/1e6,1)) for k in ("grad","jprod","jtprod","hprod")})
PY
