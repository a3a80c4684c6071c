#!/bin/bash
# round 2, GPU session 5: size fall-off A/Bs at 4e6, pandemic 5000x100 tile/LDS A/Bs, C-ABI sharded bench, profile of the bench
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_s5
mkdir -p $O
cd $R
echo "== C-ABI sharded bench paths"
timeout -k 10 300 python3 bench.py --gpus 2 --dist-backend gloo --same-device --steps 50 --warmup 10 > $O/bench_n2_gloo.json 2> $O/bench_n2_gloo.err || { echo FAIL n2; tail -5 $O/bench_n2_gloo.err; }
timeout -k 10 200 python3 bench.py --emulate-shard 3/8 --steps 200 --warmup 20 --no-cpu-baseline > $O/shard_3_8.json 2>> $O/shard.err || echo FAIL shard
echo "== 4e6 A/B"
for v in "" "wide_stores=1" "nt_stores=0" "lds_slots=48" "lds_slots=16" "block=256" "block=1024"; do
  tag=${v:-default}
  timeout -k 10 300 python3 bench.py --supports 4000000 --steps 40 --warmup 5 --no-cpu-baseline ${v:+--opt $v} > $O/b4e6_$tag.json 2>>$O/b4e6.err || echo "fail $tag"
done
echo "== pandemic A/B"
for v in "" "lds_slots=20" "lds_slots=16" "lds_slots=32" "block=256" "block=256 lds_slots=48"; do
  tag=$(echo ${v:-default} | tr ' =' '__')
  opts=""; for kv in $v; do opts="$opts --opt $kv"; done
  timeout -k 10 200 python3 tools/eval_loop.py --workload pandemic $opts > $O/pand_$tag.json 2>>$O/pand.err || echo "fail $tag"
done
timeout -k 10 200 python3 tools/eval_loop.py --workload pandemic --nt 4950 > $O/pand_nt4960.json 2>>$O/pand.err
timeout -k 10 200 python3 tools/eval_loop.py --workload pandemic --nt 4950 --opt lds_slots=20 > $O/pand_nt4960_lds20.json 2>>$O/pand.err
echo "== quadrotor cons A/B"
for v in "" "lds_slots=48" "lds_slots=12" "block=256"; do
  tag=${v:-default}
  timeout -k 10 200 python3 tools/eval_loop.py --workload quadrotor --supports 1000000 ${v:+--opt $v} > $O/quad_$tag.json 2>>$O/quad.err || echo "fail $tag"
done
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        if "ms" in j: print(os.path.basename(f), {k: round(v*1e3,2) for k,v in j["ms"].items()}, {k: round(v) for k,v in j["GBps"].items()})
        else:
            r=j["roofline"]; print(os.path.basename(f), "value %.0f ms/step %.4f jac %.4f hess %.4f pair_frac %.3f"%(j["value"], j["ms_per_step"], r["jac_ms"], r["hess_ms"], r["pair_frac"]), j.get("weak",{}).get("value"))
    except Exception as e: print(f, "ERR", e)
PY
echo "== profile of the bench"
bash tools/profile_gpu.sh > $O/profile_gpu.log 2>&1; tail -5 $O/profile_gpu.log
cp $R/gpurun_out/prof/summary.json $O/rocprof_summary.json 2>/dev/null
find $R/gpurun_out/prof/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_bench.csv
rm -rf $R/gpurun_out/prof
