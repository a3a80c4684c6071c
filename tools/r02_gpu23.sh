#!/bin/bash
# tile size vs grid size between one and two workgroups per CU
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_s23
mkdir -p $O
cd $R
for S in 140000 170000 200000 230000 262000 300000 350000 400000; do
  IEM_AB_SUPPORTS=$S timeout -k 10 280 python3 tools/ab_inproc.py "autotune=0" "autotune=0,block=1024" > $O/ab_$S.txt 2>$O/ab_$S.err || echo "fail $S"
  echo "## $S"; grep "round 2" $O/ab_$S.txt
done
