#!/bin/bash
# The five-call evaluation loops behind profiles/rNN_eval_loops.json (BASELINE's configs 3, 4, 5, the reference's own
# benchmark sizes — ESCAPE34/run_cases_gpu.jl:89-102 — and the headline size), from one tree; then a rocprofv3 kernel trace
# of the quadrotor 16 000 loop.   tools/sessions/eval_loops.sh [name]
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-eval_loops}; rm -rf $O; mkdir -p $O
cd $R
i=0
for spec in "--workload farmer --supports 100000" "--workload opf --supports 10000" "--workload quadrotor --supports 16000" "--workload quadrotor --supports 1000" "--workload pandemic" "--workload pandemic --nt 90 --nxi 128" "--workload quadrotor --supports 1000000"; do
  i=$((i+1))
  timeout -k 10 400 python3 tools/eval_loop.py $spec > $O/loop_$i.json 2> $O/loop_$i.err || { echo "FAILED $spec"; tail -3 $O/loop_$i.err; exit 1; }
  python3 - $O/loop_$i.json <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
f = j["loop_forms_ms"]
print(j["workload"], "| five calls %.1f us, deferred+pair %.1f us, TWO PHASE LAUNCHES %.1f us (%.2f of peak), ONE LAUNCH %.1f us (%.2f) | trial %.1f accepted %.1f all %.1f us" % (
    f["five_calls"] * 1e3, f["obj_deferred_fused_pair"] * 1e3, f["two_phase_launches"] * 1e3, j["loop_frac_of_8TBps"]["two_phase_launches"], f["one_launch"] * 1e3, j["loop_frac_of_8TBps"]["one_launch"], j["ms"]["eval_trial"] * 1e3, j["ms"]["eval_accepted"] * 1e3, j["ms"]["eval_all"] * 1e3))
PY
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_q16000 -- python3 $R/tools/eval_loop.py --workload quadrotor --supports 16000 > $O/prof_q16000.log 2>&1 || echo "rocprof failed"
echo ok
