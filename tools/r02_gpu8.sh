#!/bin/bash
# round 2, GPU session 8: comm tests after the multi-workgroup all-reduce; lds_slots 24 vs 48 per kernel across models (in-process)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_s8
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_comm.py -x -q -m gpu > $O/pytest_comm.log 2>&1; echo "pytest comm rc=$?"; tail -3 $O/pytest_comm.log
timeout -k 10 300 python3 tools/eval_loop_dist.py --gpus 2 --dist-backend gloo --same-device --workload pandemic --nt 4990 --supports 12 --allreduce own > $O/dist_pandemic_n2_own.json 2>>$O/dist.err || echo "fail pandemic 2"
python3 -c "
import json; j=json.loads(open('$O/dist_pandemic_n2_own.json').read().strip().splitlines()[-1]); print('pandemic n2', {k: round(v*1e3,2) for k,v in j['ms'].items()}, j['collective'])"
for wl in "opf 1000000" "farmer 1000000" "pandemic 500000" "quadrotor_oc3 500000" "quadrotor 2000000" "quadrotor 500000"; do
  set -- $wl
  IEM_AB_WORKLOAD=$1 IEM_AB_SUPPORTS=$2 timeout -k 10 280 python3 tools/ab_inproc.py "lds_slots=24" "lds_slots=48" "lds_slots=36" > $O/ab_$1_$2.txt 2>$O/ab_$1_$2.err || echo "fail $wl"
  echo "## $wl"; grep "round 2" $O/ab_$1_$2.txt
done
