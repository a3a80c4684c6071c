#!/bin/bash
# iem_tune at bench set-up: driver-form runs (--steps 20 --warmup 5), default runs, tuner test
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_s27
mkdir -p $O
cd $R
timeout -k 10 200 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tuner" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -1 $O/pytest.log
for i in 1 2 3; do
  timeout -k 10 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/driver_$i.json 2> $O/driver_$i.err
  IEM_TUNER_LOG=1 timeout -k 10 200 python3 bench.py --no-cpu-baseline > $O/bench_$i.json 2> $O/bench_$i.err
  for f in driver_$i bench_$i; do python3 -c "
import json; j=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); r=j['roofline']; print('$f', round(j['value'],1), round(j['ms_per_step'],5), round(r['jac_ms'],5), round(r['hess_ms'],5), round(r['frac'],4), j['config']['store_batch_tuner']['jac'], j['config']['store_batch_tuner']['hess'])"; done
done
