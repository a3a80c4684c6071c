#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_s20
mkdir -p $O
cd $R
for S in 1000000 4000000; do
  IEM_AB_SUPPORTS=$S timeout -k 10 280 python3 tools/ab_inproc.py "autotune=0" "autotune=0,lds_slots=48" "autotune=1" > $O/ab_$S.txt 2>$O/ab_$S.err || echo "fail $S"
  echo "## $S"; grep "round [12]" $O/ab_$S.txt
done
