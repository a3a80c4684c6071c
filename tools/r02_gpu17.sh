#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_s17
mkdir -p $O
cd $R
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -3 $O/pytest_gpu.log
timeout -k 10 200 python3 tools/eval_loop.py --workload pandemic > $O/pand.json 2>>$O/pand.err
timeout -k 10 200 python3 tools/eval_loop.py --workload quadrotor --supports 16000 > $O/quad16k.json 2>>$O/pand.err
timeout -k 10 200 python3 tools/eval_loop.py --workload opf --supports 10000 > $O/opf1e4.json 2>>$O/pand.err
timeout -k 10 200 python3 tools/eval_loop.py --workload farmer --supports 100000 > $O/farmer1e5.json 2>>$O/pand.err
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), {k: round(v*1e3,2) for k,v in j["ms"].items()}, "loop", round(j["loop_ms"]*1e3,1), "graph", round(j.get("graph_loop_ms",0)*1e3,1), "build_s", round(j["build_s"],2))
PY
