#!/bin/bash
# matrix-free products at the named sizes: event-timed per call, algorithmic bytes, rate
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_s25
mkdir -p $O
cd $R
timeout -k 10 200 python3 tools/eval_loop.py --workload quadrotor --supports 1000000 --products > $O/quadrotor_1e6.json 2>$O/q.err || echo fail q
timeout -k 10 200 python3 tools/eval_loop.py --workload pandemic --products > $O/pandemic_5e5.json 2>$O/p.err || echo fail p
timeout -k 10 200 python3 tools/eval_loop.py --workload opf --supports 1000000 --products > $O/opf_1e6.json 2>$O/o.err || echo fail o
timeout -k 10 200 python3 tools/eval_loop.py --workload farmer --supports 1000000 --products > $O/farmer_1e6.json 2>$O/f.err || echo fail f
python3 - <<PY
import json
for n in ("quadrotor_1e6","pandemic_5e5","opf_1e6","farmer_1e6"):
    j=json.loads(open("$O/%s.json"%n).read().strip().splitlines()[-1])
    print(n, {k:(round(j["ms"][k]*1e3,1), round(j["alg_bytes"][k]/1e6,1), round(j["GBps"][k]/8000,3)) for k in j["ms"]})
PY
