#!/bin/bash
# eval loops at the reference's ladder sizes under different `split_small` thresholds (side-by-side templates vs lane-fused bodies)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-split_ab}; rm -rf $O; mkdir -p $O
cd $R
for spec in "quadrotor 16000" "quadrotor 4000" "quadrotor 1000" "opf 10000" "opf 1000"; do
  set -- $spec
  for ss in 64 24 8 0; do
    timeout -k 10 300 python3 tools/eval_loop.py --workload $1 --supports $2 --opt split_small=$ss > $O/$1_$2_ss$ss.json 2> $O/$1_$2_ss$ss.err || { echo FAILED $spec $ss; tail -3 $O/$1_$2_ss$ss.err; continue; }
    python3 - $O/$1_$2_ss$ss.json $ss <<'PY'
import json, sys
j = json.load(open(sys.argv[1])); f = j["loop_forms_ms"]; m = j["ms"]
print("%-28s split_small %2s | five %.1f  pair+defer %.1f  two-phase %.1f  one %.1f us | per call: obj %.1f grad %.1f cons %.1f jac %.1f hess %.1f accepted %.1f" % (
  j["workload"][:28], sys.argv[2], f["five_calls"]*1e3, f["obj_deferred_fused_pair"]*1e3, f["two_phase_launches"]*1e3, f["one_launch"]*1e3,
  m["obj"]*1e3, m["grad"]*1e3, m["cons"]*1e3, m["jac_coord"]*1e3, m["hess_coord"]*1e3, m["eval_accepted"]*1e3))
PY
  done
done
