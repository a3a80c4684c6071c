#!/bin/bash
# Copy what tools/sessions/final_round.sh left under gpurun_out/ into profiles/ (the tracked copies the bench line and DESIGN.md cite).
set -e
cd "$(dirname "$0")/.."
P=gpurun_out/prof; F=gpurun_out/r03_final
cp $P/summary.json profiles/pmc_quadrotor_1e6.json
cp "$(find $P/stats -name '*kernel_stats.csv' | head -1)" profiles/r03_kernel_stats_bench_1e6.csv
cp "$(find $F/shard_stats -name '*kernel_stats.csv' | head -1)" profiles/r03_kernel_stats_shard_3_of_8.csv
cp $F/shard_bench.json profiles/r03_bench_shard_3_of_8_under_rocprof.json
cp $F/bench_runs.txt profiles/r03_bench_runs.txt
cp $F/bench_default.json profiles/r03_bench_default.json
python3 - <<'PY'
import json, sys
sys.path.insert(0, ".")
import bench
p = json.load(open("profiles/pmc_quadrotor_1e6.json"))
print("profile fingerprint", p.get("csrc_fingerprint"), "commit", p.get("commit"), "| tree fingerprint", bench.csrc_fingerprint())
assert p.get("csrc_fingerprint") == bench.csrc_fingerprint(), "the PMC profile is not from this csrc tree"
j = json.loads(open("profiles/r03_bench_default.json").read().strip().splitlines()[-1])
print("default bench:", j["value"], j["unit"], "roofline", {k: j["roofline"].get(k) for k in ("kernel", "frac", "frac_cold_inputs", "traffic", "traffic_stale")})
PY
