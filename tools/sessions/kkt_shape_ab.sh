#!/bin/bash
# A/B of the kkt_eliminate workgroup shape (waves per workgroup x register budget) on one workload:
#   kkt_shape_ab.sh <workload> <supports> "<wmax>:<wpe> <wmax>:<wpe> ..." ["#define ..."]     (wpe 0: the compiler's choice)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/kkt_shape_ab; mkdir -p $O
cd $R
for v in $3; do
  w=${v%%:*}; e=${v##*:}
  IEM_KKT_EXPERIMENTS=1 IEM_KKT_WMAX=$w IEM_KKT_WPE=$e IEM_KKT_DEFS="$4" timeout -k 10 250 python3 tools/kkt_chain_bench.py --workload $1 --supports $2 --cabi 0 > $O/$1_$2_w${w}_e${e}.log 2>&1 || { echo "FAILED $v"; tail -3 $O/$1_$2_w${w}_e${e}.log; exit 1; }
  echo "$1 $2 wmax=$w wpe=$e $4 $(tail -1 $O/$1_$2_w${w}_e${e}.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('factor_ms', round(d['ms']['factor'],3), 'solve_ms', round(d['ms']['solve_no_refinement'],3), 'resid', d['rel_residual_after_refinement'])")"
done
