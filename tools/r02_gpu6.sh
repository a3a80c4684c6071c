#!/bin/bash
# round 2, GPU session 6: own all-reduce vs RCCL, pandemic after auto block, whole GPU suite
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_s6
mkdir -p $O
cd $R
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -4 $O/pytest_gpu.log
echo "== collective: world 1 (nccl process group), own vs rccl"
for w in farmer opf; do
  S=100000; [ $w = opf ] && S=10000
  for ar in own rccl; do
    timeout -k 10 200 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 tools/eval_loop_dist.py --workload $w --supports $S --allreduce $ar > $O/dist_${w}_n1_$ar.json 2>>$O/dist.err || echo "fail $w $ar"
  done
done
echo "== collective: 2 and 4 ranks sharing the GPU (gloo for handles), own"
for n in 2 4; do
  timeout -k 10 300 python3 tools/eval_loop_dist.py --gpus $n --dist-backend gloo --same-device --workload farmer --supports 25000 --allreduce own > $O/dist_farmer_n${n}_own.json 2>>$O/dist.err || echo "fail farmer $n"
  timeout -k 10 300 python3 tools/eval_loop_dist.py --gpus $n --dist-backend gloo --same-device --workload opf --supports 2500 --allreduce own > $O/dist_opf_n${n}_own.json 2>>$O/dist.err || echo "fail opf $n"
  timeout -k 10 300 python3 tools/eval_loop_dist.py --gpus $n --dist-backend gloo --same-device --workload quadrotor --supports 125000 --allreduce own > $O/dist_quadrotor_n${n}_own.json 2>>$O/dist.err || echo "fail quad $n"
  timeout -k 10 300 python3 tools/eval_loop_dist.py --gpus $n --dist-backend gloo --same-device --workload pandemic --nt 4990 --supports 12 --allreduce own > $O/dist_pandemic_n${n}_own.json 2>>$O/dist.err || echo "fail pandemic $n"
done
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/dist_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), "loop_us %.1f"%(j["loop_ms"]*1e3), {k: round(v*1e3,2) for k,v in j["ms"].items()}, j["collective"])
    except Exception as e: print(f, "ERR", e)
PY
echo "== pandemic after auto block, + block 256 lds variants"
for v in "" "lds_slots=12" "lds_slots=16" "lds_slots=8" "block=512"; do
  tag=$(echo ${v:-auto} | tr ' =' '__')
  timeout -k 10 200 python3 tools/eval_loop.py --workload pandemic ${v:+--opt $v} > $O/pand_$tag.json 2>>$O/pand.err || echo "fail $tag"
done
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/pand_*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), {k: round(v*1e3,2) for k,v in j["ms"].items()}, {k: round(v) for k,v in j["GBps"].items() if k in ("cons","jac_coord","hess_coord")})
PY
