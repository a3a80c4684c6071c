#!/bin/bash
# store-batch tuner, 20 samples / medians: decisions in bench runs (log), in-process A/B at 1e6 and 4e6, its test
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_s21
mkdir -p $O
cd $R
timeout -k 10 200 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tuner" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -1 $O/pytest.log
for i in 1 2 3 4; do
  IEM_TUNER_LOG=1 timeout -k 10 200 python3 bench.py --no-weak > $O/bench_$i.json 2> $O/bench_$i.err
  python3 -c "
import json; j=json.loads(open('$O/bench_$i.json').read().strip().splitlines()[-1]); r=j['roofline']; print('bench', j['value'], j['ms_per_step'], r['jac_ms'], r['hess_ms'], r['frac'])"
  grep "iem tuner" $O/bench_$i.err | sed 's/.*medians/  medians/'
done
for S in 1000000 4000000; do
  IEM_AB_SUPPORTS=$S timeout -k 10 280 python3 tools/ab_inproc.py "autotune=0" "autotune=0,lds_slots=48" "autotune=1" > $O/ab_$S.txt 2>$O/ab_$S.err || echo "fail $S"
  echo "## $S"; grep "round [12]" $O/ab_$S.txt
done
