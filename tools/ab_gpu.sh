#!/bin/bash
# A/B matrix of generator knobs at the headline size (run through gpurun); every
# variant runs in the same box, back to back, so the comparison is box-internal.
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/ab
i=0
for cfg in "$@"; do
  i=$((i+1))
  python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline $cfg > $R/gpurun_out/ab/run$i.log 2>&1
  python3 - <<PY
import json
l=open("$R/gpurun_out/ab/run$i.log").read().strip().splitlines()[-1]
try:
    d=json.loads(l); r=d["roofline"]; print("%-40s ms/step %.4f jac %.4f hess %.4f pair_frac %.3f"%("$cfg",d["ms_per_step"],r["jac_ms"],r["hess_ms"],r["pair_frac"]))
except Exception as e: print("FAILED $cfg", l[-300:])
PY
done
