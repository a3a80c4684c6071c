#!/bin/bash
# round 2, GPU session 10: 32-bit scalar-compare flush loops — correctness (whole GPU suite) and effect at every size
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_s10
mkdir -p $O
cd $R
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -4 $O/pytest_gpu.log
for r in 3; do
  timeout -k 10 200 python3 bench.py --emulate-shard $r/8 --steps 200 --warmup 20 --no-cpu-baseline > $O/shard_${r}_8.json 2>> $O/shard.err || echo FAIL shard
  timeout -k 10 200 python3 bench.py --emulate-shard 1/4 --steps 200 --warmup 20 --no-cpu-baseline > $O/shard_1_4.json 2>> $O/shard.err || echo FAIL shard
done
timeout -k 10 300 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_n1.json 2> $O/bench_n1.err || echo FAIL n1
for wl in "quadrotor 16000" "quadrotor 1000000" "opf 16000" "opf 1000000" "farmer 100000"; do
  set -- $wl
  timeout -k 10 200 python3 tools/eval_loop.py --workload $1 --supports $2 > $O/loop_$1_$2.json 2>>$O/loop.err || echo "fail $wl"
done
for v in "flat2d=1" "flat2d=0"; do
  timeout -k 10 200 python3 tools/eval_loop.py --workload pandemic --opt $v > $O/pand_$v.json 2>>$O/pand.err || echo "fail $v"
done
IEM_AB_WORKLOAD=pandemic IEM_AB_SUPPORTS=500000 timeout -k 10 280 python3 tools/ab_inproc.py "flat2d=0" "flat2d=1" > $O/ab_pandemic.txt 2>$O/ab_pandemic.err || echo "fail ab"
grep "round 2" $O/ab_pandemic.txt
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        if "ms" in j: print(os.path.basename(f), {k: round(v*1e3,2) for k,v in j["ms"].items()}, "loop", round(j["loop_ms"]*1e3,1), "graph", round(j.get("graph_loop_ms",0)*1e3,1))
        else:
            r=j["roofline"]; print(os.path.basename(f), "value %.0f ms/step %.4f jac %.4f hess %.4f pair_frac %.3f"%(j["value"], j["ms_per_step"], r["jac_ms"], r["hess_ms"], r["pair_frac"]))
    except Exception as e: print(f, "ERR", e)
PY
