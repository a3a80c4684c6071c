#!/bin/bash
# Copy what tools/sessions/final_round.sh left under gpurun_out/ into profiles/ (the tracked copies the bench line and DESIGN.md cite).
set -e
cd "$(dirname "$0")/.."
R=${IEM_ROUND:-r04}
P=gpurun_out/prof; F=gpurun_out/${R}_final
cp $P/summary.json profiles/pmc_quadrotor_1e6.json
cp "$(ls -t $(find $P/stats -name '*kernel_stats.csv') | head -1)" profiles/${R}_kernel_stats_bench_1e6.csv     # the NEWEST: gpurun_out/ accumulates earlier sessions' files
cp "$(ls -t $(find $F/shard_stats -name '*kernel_stats.csv') | head -1)" profiles/${R}_kernel_stats_shard_3_of_8.csv
cp $F/shard_bench.json profiles/${R}_bench_shard_3_of_8_under_rocprof.json
cp $F/bench_runs.txt profiles/${R}_bench_runs.txt
cp $F/bench_default.json profiles/${R}_bench_default.json
cp $F/credit_stalls.txt profiles/${R}_jac_credit_stalls.txt 2>/dev/null || true
python3 - <<'PY'
import json, sys
sys.path.insert(0, ".")
import bench
p = json.load(open("profiles/pmc_quadrotor_1e6.json"))
print("profile fingerprint", p.get("csrc_fingerprint"), "commit", p.get("commit"), "| tree fingerprint", bench.csrc_fingerprint())
assert p.get("csrc_fingerprint") == bench.csrc_fingerprint(), "the PMC profile is not from this csrc tree"
import os
j = json.loads(open("profiles/" + os.environ.get("IEM_ROUND", "r04") + "_bench_default.json").read().strip().splitlines()[-1])
print("default bench:", j["value"], j["unit"], "roofline", {k: j["roofline"].get(k) for k in ("kernel", "frac", "frac_cold_inputs", "traffic", "traffic_stale")})
PY
