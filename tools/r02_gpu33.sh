#!/bin/bash
# occupancy cliff of cons! (86 VGPRs: 2 workgroups per CU): __launch_bounds__(512, 3)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_s33
mkdir -p $O
cd $R
for mw in 0 3; do
  for w in "quadrotor 1000000" "opf 1000000" "farmer 1000000"; do
    set -- $w
    timeout -k 10 250 python3 tools/eval_loop.py --workload $1 --supports $2 --products --opt min_waves=$mw > $O/$1_mw$mw.json 2>$O/$1_mw$mw.err || echo fail $1 $mw
  done
  timeout -k 10 250 python3 tools/eval_loop.py --workload pandemic --products --opt min_waves=$mw > $O/pandemic_mw$mw.json 2>$O/pandemic_mw$mw.err || echo fail pandemic
done
python3 - <<PY
import json
for n in ("quadrotor","opf","farmer","pandemic"):
    for mw in (0,3):
        j=json.loads(open("$O/%s_mw%d.json"%(n,mw)).read().strip().splitlines()[-1])
        print(n, "min_waves", mw, {k:round(j["ms"][k]*1e3,1) for k in j["ms"]})
PY
