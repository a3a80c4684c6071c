#!/bin/bash
# round 2, GPU session 7: in-process knob A/B (same output buffers for every variant) at four sizes
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_s7
mkdir -p $O
cd $R
for S in 4000000 1000000 250000 125000; do
  IEM_AB_SUPPORTS=$S timeout -k 10 280 python3 tools/ab_inproc.py "lds_slots=24" "block=256" "lds_slots=48" "lds_slots=32" "lds_slots=16" "block=256,lds_slots=48" > $O/ab_$S.txt 2>$O/ab_$S.err || echo "fail $S"
  grep "round 2" $O/ab_$S.txt
done
