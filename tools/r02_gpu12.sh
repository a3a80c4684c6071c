#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_s12
mkdir -p $O
cd $R
for wl in "quadrotor 125000" "quadrotor 1000000" "quadrotor 16000" "opf 16000"; do
  set -- $wl
  IEM_AB_WORKLOAD=$1 IEM_AB_SUPPORTS=$2 timeout -k 10 280 python3 tools/ab_inproc.py "flush32=0" "flush32=2" "flush32=1" > $O/ab_$1_$2.txt 2>$O/ab_$1_$2.err || echo "fail $wl"
  echo "## $wl"; grep "round [12]" $O/ab_$1_$2.txt
done
IEM_AB_WORKLOAD=pandemic IEM_AB_SUPPORTS=500000 timeout -k 10 280 python3 tools/ab_inproc.py "flush32=0,flat2d=0" "flush32=2,flat2d=0" "flush32=2,flat2d=1" > $O/ab_pandemic.txt 2>$O/ab_pandemic.err || echo "fail pandemic"
echo "## pandemic"; grep "round [12]" $O/ab_pandemic.txt
