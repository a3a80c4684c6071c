#!/bin/bash
# round 2, GPU session 19: store-batch tuner — correctness test, in-process A/B, bench
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_s19
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tuner or offline or two_handles or 1e6" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
for S in 1000000 4000000 500000; do
  IEM_AB_SUPPORTS=$S timeout -k 10 280 python3 tools/ab_inproc.py "autotune=0" "autotune=1" > $O/ab_$S.txt 2>$O/ab_$S.err || echo "fail $S"
  echo "## $S"; grep "round [12]" $O/ab_$S.txt
done
IEM_AB_WORKLOAD=opf IEM_AB_SUPPORTS=1000000 timeout -k 10 280 python3 tools/ab_inproc.py "autotune=0" "autotune=1" > $O/ab_opf.txt 2>$O/ab_opf.err; echo "## opf 1e6"; grep "round [12]" $O/ab_opf.txt
for i in 1 2 3; do
  timeout -k 10 300 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_$i.json 2>>$O/bench.err || echo FAIL bench
  timeout -k 10 300 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --opt autotune=0 > $O/bench_notune_$i.json 2>>$O/bench.err || echo FAIL bench
done
timeout -k 10 300 python3 bench.py --supports 4000000 --steps 40 --warmup 10 --no-cpu-baseline > $O/bench_4e6.json 2>>$O/bench.err
timeout -k 10 300 python3 bench.py --supports 4000000 --steps 40 --warmup 10 --no-cpu-baseline --opt autotune=0 > $O/bench_4e6_notune.json 2>>$O/bench.err
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/bench*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); r=j["roofline"]; print(os.path.basename(f), "%.0f"%j["value"], "ms/step %.4f"%j["ms_per_step"], "jac %.4f hess %.4f pair_frac %.3f"%(r["jac_ms"], r["hess_ms"], r["pair_frac"]), j["config"]["kernels_from"])
PY
