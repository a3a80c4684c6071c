#!/bin/bash
# kkt_eliminate's workgroup shape for the 20 x 20 blocks of the hub solver (pandemic 5 000 x 100): waves per block x register budget
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04_hub_shape}; mkdir -p $O
export IEM_KKT_EXPERIMENTS=1
for cfg in "default" "1 4" "1 6" "1 8" "2 6" "2 8" "1 0"; do
  set -- $cfg
  if [ "$1" = default ]; then unset IEM_KKT_WMAX IEM_KKT_WPE; else export IEM_KKT_WMAX=$1 IEM_KKT_WPE=$2; fi
  timeout -k 10 200 python3 $R/tools/kkt_hub_bench.py --iters 3 > $O/run.json 2> $O/run.err || { tail -3 $O/run.err; exit 1; }
  python3 - "$cfg" $O/run.json <<'PY' | tee -a $O/table.txt
import json, sys
j = json.load(open(sys.argv[2])); p = j["factor_phases_ms_synchronised"]
print(f"wmax/wpe {sys.argv[1]:8s} factor {j['ms']['factor']:6.2f} solve {j['ms']['solve']:6.2f}  eliminate {p['chain eliminate']:5.2f}  ldl {p[chr(100)+'ense block LDL'+chr(39)+' of the hubs']:5.2f}  resid {j['abs_residual']['one_refinement']:.1e} inertia {j['inertia']}")
PY
done
