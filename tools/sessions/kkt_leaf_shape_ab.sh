R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_leaf; mkdir -p $O
export IEM_KKT_EXPERIMENTS=1
for cfg in "2 2" "3 2" "6 2" "6 0" "3 0"; do
  set -- $cfg
  export IEM_KKT_WMAX=$1 IEM_KKT_WPE=$2
  timeout -k 10 200 python3 $R/tools/kkt_hub_bench.py --iters 3 > $O/run.json 2> $O/run.err || { tail -3 $O/run.err; exit 1; }
  python3 -c "
import json,sys
j=json.load(open('$O/run.json')); p=j['factor_phases_ms_synchronised']
print('wmax/wpe $cfg', 'factor %.2f' % j['ms']['factor'], 'ldl %.2f' % [v for k,v in p.items() if 'LDL' in k][0], 'resid %.1e' % j['abs_residual']['one_refinement'])" | tee -a $O/table.txt
done
