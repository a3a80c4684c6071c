#!/bin/bash
# round 2: final validation of the committed state — what the round driver runs, plus the default bench
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_final
mkdir -p $O
cd $R
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -2 $O/pytest_gpu.log
timeout -k 10 600 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c1-420 $O/bench_default.json
timeout -k 10 300 python3 bench.py --gpus 2 --dist-backend gloo --same-device --steps 50 --warmup 10 > $O/bench_n2_gloo.json 2> $O/bench_n2_gloo.err; echo "bench n2 rc=$?"
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err; echo "bench driver-form rc=$?"; python3 -c "
import json; j=json.loads(open('$O/bench_driver_form.json').read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['kernel'], j['cpu_baseline']['value'], j['config']['kernels_from'])"
