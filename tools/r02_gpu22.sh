#!/bin/bash
# shard-size A/B (what one of 8 / 4 / 2 GPUs runs in the strong-scaling bench): tile size, LDS batch, side-by-side bodies
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_s22
mkdir -p $O
cd $R
for S in 125000 250000 500000; do
  IEM_AB_SUPPORTS=$S timeout -k 10 280 python3 tools/ab_inproc.py "autotune=0" "autotune=0,block=256" "autotune=0,block=1024" "autotune=0,lds_slots=16" "autotune=0,lds_slots=48" "autotune=0,split_small=1100" > $O/ab_$S.txt 2>$O/ab_$S.err || echo "fail $S"
  echo "## $S"; grep "round [12]" $O/ab_$S.txt
done
