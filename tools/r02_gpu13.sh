#!/bin/bash
# round 2, GPU session 13: validation of the round's defaults — smoke, whole GPU suite, soak seeds, > 2^31 entries, default bench
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_s13
mkdir -p $O
cd $R
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -3 $O/pytest_gpu.log
IEM_EXTRA_SEEDS="200:240" IEM_EXTRA_BIG_SEEDS="300:312" timeout -k 10 1000 python3 -m pytest tests/test_random_templates.py -x -q -m gpu > $O/pytest_soak.log 2>&1; echo "soak rc=$?"; tail -2 $O/pytest_soak.log
timeout -k 10 600 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c1-900 $O/bench_default.json
timeout -k 10 1100 python3 tools/huge_check.py > $O/huge.log 2>&1; echo "huge rc=$?"; tail -3 $O/huge.log
