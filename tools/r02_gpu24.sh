#!/bin/bash
# LDS store-batch size at 1e6 and 2e6 supports (placement-insensitive alternative for the tuner)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_s24
mkdir -p $O
cd $R
for S in 1000000 2000000; do
  IEM_AB_SUPPORTS=$S timeout -k 10 280 python3 tools/ab_inproc.py "autotune=0" "autotune=0,lds_slots=40" "autotune=0,lds_slots=48" "autotune=0,lds_slots=56" "autotune=0,lds_slots=64" "autotune=0,lds_slots=72" > $O/ab_$S.txt 2>$O/ab_$S.err || echo "fail $S"
  echo "## $S"; grep "round [12]" $O/ab_$S.txt
done
