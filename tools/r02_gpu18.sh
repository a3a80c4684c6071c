#!/bin/bash
# round 2, GPU session 18: BASELINE configs 4/5 — one GPU at the named size vs one 1/8 shard of it (what each of 8 GPUs would run)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_s18
mkdir -p $O
cd $R
timeout -k 10 300 python3 -m pytest tests/test_gpu_comm.py -x -q -m gpu > $O/pytest_comm.log 2>&1; echo "pytest comm rc=$?"; tail -2 $O/pytest_comm.log
for wl in "opf 10000" "opf 1250" "farmer 100000" "farmer 12500" "opf 1000000" "opf 125000" "farmer 1000000" "farmer 125000"; do
  set -- $wl
  timeout -k 10 200 python3 tools/eval_loop.py --workload $1 --supports $2 > $O/loop_$1_$2.json 2>>$O/loop.err || echo "fail $wl"
done
# the distributed loop itself (weak: per-rank size fixed), world 1 and 2 on the one GPU
for wl in "opf 1250" "farmer 12500"; do
  set -- $wl
  timeout -k 10 200 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 tools/eval_loop_dist.py --workload $1 --supports $2 > $O/dist_$1_$2_n1.json 2>>$O/dist.err || echo "fail dist $wl"
  timeout -k 10 300 python3 tools/eval_loop_dist.py --gpus 2 --dist-backend gloo --same-device --workload $1 --supports $2 > $O/dist_$1_$2_n2.json 2>>$O/dist.err || echo "fail dist2 $wl"
done
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), {k: round(v*1e3,2) for k,v in j["ms"].items()}, "loop us", round(j["loop_ms"]*1e3,1), "graph", round(j.get("graph_loop_ms",0)*1e3,1))
PY
