#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_s15
mkdir -p $O
cd $R
for sh in 0/8 3/8 7/8 1/2; do
  timeout -k 10 200 python3 bench.py --emulate-shard $sh --steps 100 --warmup 10 --no-cpu-baseline > $O/shard_$(echo $sh | tr / _).json 2>> $O/shard.err || echo FAIL shard $sh
done
timeout -k 10 300 python3 bench.py --gpus 2 --dist-backend gloo --same-device --steps 50 --warmup 10 > $O/bench_n2_gloo.json 2> $O/bench_n2_gloo.err || { echo FAIL n2; tail -5 $O/bench_n2_gloo.err; }
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), "%.0f"%j["value"], "%.4f"%j["ms_per_step"], j["config"]["kernels_from"], j.get("weak",{}).get("value"))
PY
