#!/bin/bash
# lazy loads in the product kernels: OPF / quadrotor / pandemic / farmer, lazy_loads 0 / 1 / 2 (lazy_min_loads 8 so that every model takes part)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_s29
mkdir -p $O
cd $R
for ll in 0 1 2; do
  for w in "opf 1000000" "quadrotor 1000000" "farmer 1000000" "hovercraft 1000000"; do
    set -- $w
    timeout -k 10 250 python3 tools/eval_loop.py --workload $1 --supports $2 --products --opt lazy_loads=$ll --opt lazy_min_loads=8 > $O/$1_ll$ll.json 2>$O/$1_ll$ll.err || echo fail $1 $ll
  done
  timeout -k 10 250 python3 tools/eval_loop.py --workload pandemic --products --opt lazy_loads=$ll --opt lazy_min_loads=8 > $O/pandemic_ll$ll.json 2>$O/pandemic_ll$ll.err || echo fail pandemic $ll
done
python3 - <<PY
import json
for n in ("opf","quadrotor","farmer","hovercraft","pandemic"):
    for ll in (0,1,2):
        j=json.loads(open("$O/%s_ll%d.json"%(n,ll)).read().strip().splitlines()[-1])
        print(n, "lazy", ll, {k:round(j["ms"][k]*1e3,1) for k in ("grad","jprod","jtprod","hprod")})
PY
