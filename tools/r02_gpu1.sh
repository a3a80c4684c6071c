#!/bin/bash
# round 2, GPU session 1: N>1 launch rehearsal, strong-scaling shard proxies, block-size sweep at shard sizes,
# baseline obj profile, counter list
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_s1
mkdir -p $O
cd $R
echo "== plain --gpus 2 launch (gloo, same device)"; 
timeout -k 10 300 python3 bench.py --gpus 2 --dist-backend gloo --same-device --steps 50 --warmup 10 > $O/bench_n2_gloo.json 2> $O/bench_n2_gloo.err || { echo FAIL n2; tail -5 $O/bench_n2_gloo.err; exit 1; }
tail -c 600 $O/bench_n2_gloo.json; echo
echo "== 1-GPU baseline (no cpu baseline)"
timeout -k 10 300 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_n1.json 2> $O/bench_n1.err || { echo FAIL n1; exit 1; }
for r in 0 3 7; do
  for g in "" "--graph"; do
    timeout -k 10 200 python3 bench.py --scaling strong --emulate-shard $r/8 --steps 200 --warmup 20 --no-cpu-baseline $g > $O/shard_${r}_8${g}.json 2>> $O/shard.err || { echo FAIL shard $r; exit 1; }
  done
done
for b in 128 256 1024; do
  timeout -k 10 200 python3 bench.py --scaling strong --emulate-shard 3/8 --steps 200 --warmup 20 --no-cpu-baseline --opt block=$b > $O/shard_3_8_block$b.json 2>> $O/shard.err || echo "block $b failed"
done
for b in 256; do
  timeout -k 10 200 python3 bench.py --scaling strong --emulate-shard 1/4 --steps 200 --warmup 20 --no-cpu-baseline --opt block=$b > $O/shard_1_4_block$b.json 2>> $O/shard.err || echo "block $b failed"
  timeout -k 10 200 python3 bench.py --scaling strong --emulate-shard 1/2 --steps 200 --warmup 20 --no-cpu-baseline --opt block=$b > $O/shard_1_2_block$b.json 2>> $O/shard.err || echo "block $b failed"
done
timeout -k 10 200 python3 bench.py --scaling strong --emulate-shard 1/4 --steps 200 --warmup 20 --no-cpu-baseline > $O/shard_1_4.json 2>> $O/shard.err
timeout -k 10 200 python3 bench.py --scaling strong --emulate-shard 1/2 --steps 200 --warmup 20 --no-cpu-baseline > $O/shard_1_2.json 2>> $O/shard.err
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        r=j.get("roofline",{})
        print(os.path.basename(f), "value %.0f ms/step %.4f jac %.4f hess %.4f" % (j["value"], j["ms_per_step"], r.get("jac_ms",0), r.get("hess_ms",0)), j.get("weak",{}).get("value"))
    except Exception as e:
        print(f, "ERR", e)
PY
echo "== eval loop baseline (obj) 1e6 + ladder size"
timeout -k 10 200 python3 tools/eval_loop.py --workload quadrotor --supports 1000000 > $O/loop_quad_1e6.json 2>$O/loop.err
timeout -k 10 200 python3 tools/eval_loop.py --workload quadrotor --supports 16000 > $O/loop_quad_16000.json 2>>$O/loop.err
timeout -k 10 200 python3 tools/eval_loop.py --workload pandemic > $O/loop_pandemic_5e5.json 2>>$O/loop.err
cat $O/loop_quad_1e6.json $O/loop_quad_16000.json $O/loop_pandemic_5e5.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters.txt 2>&1 || echo "rocprofv3 -L failed"
wc -l $O/counters.txt
