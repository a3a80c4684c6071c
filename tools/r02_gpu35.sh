#!/bin/bash
# candidates for the tuner's second code object: one workgroup per CU in other shapes
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_s35
mkdir -p $O
cd $R
IEM_AB_SUPPORTS=1000000 timeout -k 10 280 python3 tools/ab_inproc.py "autotune=0" "autotune=0,lds_slots=48" "autotune=0,block=1024,lds_slots=16" "autotune=0,block=1024,lds_slots=18" "autotune=0,block=256,lds_slots=72" "autotune=0,lds_slots=48,nt_stores=0" > $O/ab.txt 2>$O/ab.err || echo "fail"
grep "round [12]" $O/ab.txt
