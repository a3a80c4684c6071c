#!/bin/bash
# End-of-round evidence from ONE tree (run through gpurun; IEM_COMMIT=$(git rev-parse --short HEAD) in the environment):
# the rocprof recipe of the default bench (kernel stats + PMC passes -> gpurun_out/prof), kernel stats of a 1/8 shard,
# ten consecutive default bench processes, the N = 2 rehearsal.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${IEM_ROUND:-r04}_final; mkdir -p $O
cd $R
bash tools/profile_gpu.sh > $O/profile.log 2>&1 || { tail -5 $O/profile.log; exit 1; }
echo "profile ok"
# write-queue credit stalls of jac_coord! in both shapes (split bodies, the default; one body): separate --pmc passes of the same command
( cd /tmp && export TMPDIR=/tmp
  for shape in 1 0; do
    rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum --output-format csv -d $O/pmc_credit_split$shape -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-cold --no-live-traffic --opt jac_split=$shape > $O/pmc_credit_split$shape.log 2>&1 || echo "credit pass $shape failed"
  done ) 
python3 - $O <<'PY' > $O/credit_stalls.txt 2>&1
import csv, glob, collections, sys, os
for shape in (1, 0):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(sys.argv[1], f"pmc_credit_split{shape}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith(("iem_jac", "iem_hess")):
                acc[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        print(f"jac_split={shape}  {k:16s} {c:40s} mean per launch {sum(v) / len(v):14.0f}  ({len(v)} launches)")
PY
cat $O/credit_stalls.txt
cp $R/gpurun_out/prof/summary.json $R/profiles/pmc_quadrotor_1e6.json    # (the box's copy of the tree: the bench lines below cite THIS profile; tools/collect_final.sh makes the same copy at home)
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/shard_stats -- python3 $R/bench.py --emulate-shard 3/8 --steps 200 --warmup 20 --no-cpu-baseline --no-cold --no-live-traffic > $O/shard_bench.json 2> $O/shard_bench.err ) || exit 1
echo "shard stats ok"
for i in 1 2 3 4 5 6 7 8 9 10; do
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-cold --no-variants --no-live-traffic > $O/run_$i.json 2> $O/run_$i.err || exit 1
  python3 - $O/run_$i.json $i <<'PY' >> $O/bench_runs.txt
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = j["roofline"]
print(f"run {sys.argv[2]:>2}: value {j['value']:.1f} pairs/s (two calls)  ms/step {j['ms_per_step']:.4f}  jac {r['jac_ms']:.4f} ({r['jac_frac']:.3f})  hess {r['hess_ms']:.4f} ({r['hess_frac']:.3f})  pair_frac {r['pair_frac']:.3f}  fused pair {j['fused_pair']['value']:.1f} pairs/s, kernel {r['pair_kernel_ms']:.4f} ms ({r['pair_kernel_frac']:.3f})")
PY
done
cat $O/bench_runs.txt
timeout -k 10 600 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1
echo "default bench ok"
