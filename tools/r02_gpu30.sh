#!/bin/bash
# lazy loads on by default (>= 48 loads): GPU suite + OPF products from the prebuilt objects
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_s30
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -2 $O/pytest_gpu.log
timeout -k 10 250 python3 tools/eval_loop.py --workload opf --supports 1000000 --products > $O/opf_1e6.json 2>$O/opf.err || echo fail opf
timeout -k 10 250 python3 tools/eval_loop.py --workload opf --supports 10000 --products > $O/opf_1e4.json 2>$O/opf4.err || echo fail opf4
python3 - <<PY
import json
for n in ("opf_1e6","opf_1e4"):
    j=json.loads(open("$O/%s.json"%n).read().strip().splitlines()[-1])
    print(n, {k:round(j["ms"][k]*1e3,1) for k in j["ms"]})
PY
