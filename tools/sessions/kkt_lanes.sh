#!/bin/bash
# chain KKT solver on the reference's pandemic ladder (ESCAPE34/run_cases_gpu.jl:99-102: (100, 8) and (100, 128)): one chain per scenario
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-kkt_lanes}; rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_kkt_cabi.py tests/test_kkt_chain.py -x -q -m gpu > $O/pytest.log 2>&1 || { tail -20 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for nxi in 8 128; do
  timeout -k 10 600 python3 tools/kkt_chain_bench.py --workload pandemic --supports 100 --nxi $nxi > $O/pandemic_100x$nxi.json 2> $O/pandemic_100x$nxi.err || { echo FAILED $nxi; tail -5 $O/pandemic_100x$nxi.err; exit 1; }
  python3 - $O/pandemic_100x$nxi.json <<'PY'
import json, sys
j = json.load(open(sys.argv[1])); c = j["c_abi_object"]
print("pandemic 110 x", j["chain"]["S"] // 110, "chain", j["chain"], "| python-held: factor %.2f ms solve+1 refinement %.2f ms residual %.1e | C-ABI object: create %.2f s assemble %.3f assemble+factor %.2f solve %.2f ms inertia %s ncon %d" % (
  j["ms"]["factor"], j["ms"]["solve_1_refinement"], j["rel_residual_after_refinement"], c["create_s"], c["assemble_ms"], c["assemble_factor_ms"], c["solve_ms"], c["inertia"], j["ncon"]))
PY
done
echo ok
