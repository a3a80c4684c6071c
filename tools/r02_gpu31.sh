#!/bin/bash
# lazy loads in the block-store kinds (cons!/jac/hess): OPF and quadrotor, in-process A/B into the same buffers + loop timings
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_s31
mkdir -p $O
cd $R
for w in opf quadrotor; do
  IEM_AB_WORKLOAD=$w IEM_AB_SUPPORTS=1000000 timeout -k 10 280 python3 tools/ab_inproc.py "autotune=0" "autotune=0,lazy_all_kinds=1,lazy_min_loads=16" "autotune=0,lazy_all_kinds=1,lazy_min_loads=16,lazy_loads=1" > $O/ab_$w.txt 2>$O/ab_$w.err || echo "fail $w"
  echo "## $w"; grep "round [12]" $O/ab_$w.txt
done
for v in "lazy_all_kinds=0" "lazy_all_kinds=1"; do
  timeout -k 10 250 python3 tools/eval_loop.py --workload opf --supports 1000000 --opt $v --opt lazy_min_loads=16 > $O/opf_$v.json 2>$O/opf_$v.err || echo fail
  python3 -c "
import json; j=json.loads(open('$O/opf_$v.json').read().strip().splitlines()[-1]); print('$v', {k:round(j['ms'][k]*1e3,1) for k in j['ms']})"
done
