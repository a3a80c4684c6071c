#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_s16
mkdir -p $O
cd $R
for sh in 0/8 3/8 7/8; do
  timeout -k 10 200 python3 bench.py --emulate-shard $sh --steps 400 --warmup 40 --no-cpu-baseline > $O/shard_$(echo $sh | tr / _).json 2>> $O/shard.err || echo FAIL shard $sh
done
timeout -k 10 200 python3 bench.py --emulate-shard 3/8 --steps 400 --warmup 40 --no-cpu-baseline --graph > $O/shard_3_8_graph.json 2>> $O/shard.err || echo FAIL graph
timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_n1.json 2>> $O/shard.err || echo FAIL n1
timeout -k 10 300 python3 bench.py --gpus 2 --dist-backend gloo --same-device --steps 50 --warmup 10 > $O/bench_n2_gloo.json 2> $O/bench_n2_gloo.err || { echo FAIL n2; tail -5 $O/bench_n2_gloo.err; }
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); r=j["roofline"]; print(os.path.basename(f), "%.0f"%j["value"], "ms/step %.4f"%j["ms_per_step"], "jac %.4f hess %.4f"%(r["jac_ms"], r["hess_ms"]), j["config"]["kernels_from"], j["config"]["launch"])
PY
