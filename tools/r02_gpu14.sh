#!/bin/bash
# round 2, GPU session 14: counter profile of the WHOLE five-call loop (obj included) at 1e6 and of a 1/8 shard
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_s14
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
pass() {  # name counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_$name -- python3 $R/tools/eval_loop.py --workload quadrotor --supports 1000000 --iters 5 > $O/pmc_$name.log 2>&1 || echo "pass $name failed"
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES
python3 - <<PY
import csv,glob,collections,json
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("iem_"): acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out={k:{c:sum(v)/len(v) for c,v in d.items()} for k,d in acc.items()}
for k,c in out.items():
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c: c["hbm_bytes_per_launch"]=(2*c["FETCH_SIZE"]+c["WRITE_SIZE"])*1024
json.dump(out, open("$O/loop_counters_quadrotor_1e6.json","w"), indent=1)
print(json.dumps({k:{n:round(v) for n,v in c.items() if n in ("hbm_bytes_per_launch","FETCH_SIZE","WRITE_SIZE","SQ_WAIT_INST_ANY","SQ_WAVE_CYCLES","SQ_WAIT_ANY","SQ_INSTS_VALU","SQ_WAVES")} for k,c in out.items()}, indent=0))
PY
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_sq
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_shard -- python3 $R/bench.py --emulate-shard 3/8 --steps 200 --warmup 20 --no-cpu-baseline > $O/st_shard.log 2>&1 || echo "rocprof shard failed"
find $O/st_shard -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_shard_3_8.csv; head -4 $O/kernel_stats_shard_3_8.csv
rm -rf $O/st_shard
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_pand -- python3 $R/tools/eval_loop.py --workload pandemic --iters 50 > $O/st_pand.log 2>&1 || echo "rocprof pand failed"
find $O/st_pand -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_pandemic_5e5.csv; head -8 $O/kernel_stats_pandemic_5e5.csv
rm -rf $O/st_pand
