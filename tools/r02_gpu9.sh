#!/bin/bash
# round 2, GPU session 9: flat 2-D grids (ordinal block store): whole GPU suite + pandemic A/B in one process
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_s9
mkdir -p $O
cd $R
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -4 $O/pytest_gpu.log
IEM_AB_WORKLOAD=pandemic IEM_AB_SUPPORTS=500000 timeout -k 10 280 python3 tools/ab_inproc.py "flat2d=0" "flat2d=1" "flat2d=1,lds_slots=20" "flat2d=0,block=512" "flat2d=1,block=256" > $O/ab_pandemic.txt 2>$O/ab_pandemic.err || echo "fail ab"
grep "round 2" $O/ab_pandemic.txt
for v in "flat2d=1" "flat2d=0"; do
  timeout -k 10 200 python3 tools/eval_loop.py --workload pandemic --opt $v > $O/pand_$v.json 2>>$O/pand.err || echo "fail $v"
done
timeout -k 10 200 python3 tools/eval_loop.py --workload pandemic --nt 90 --nxi 128 > $O/pand_100x128.json 2>>$O/pand.err
timeout -k 10 200 python3 tools/eval_loop.py --workload pandemic --nt 90 --nxi 128 --opt flat2d=0 > $O/pand_100x128_flat2d=0.json 2>>$O/pand.err
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/pand_*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), {k: round(v*1e3,2) for k,v in j["ms"].items()}, {k: round(v) for k,v in j["GBps"].items() if k in ("cons","jac_coord","hess_coord")}, "graph loop us", round(j.get("graph_loop_ms",0)*1e3,1))
PY
python3 tests/comm_worker.py 2>/dev/null | tail -1
