#!/bin/bash
# OPF product kernels spill (256 VGPRs, 150 spilled): tile size / contraction A/B on the whole loop + products
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_s26
mkdir -p $O
cd $R
for v in "autotune=0" "autotune=0 block=256" "autotune=0 det_shared=0" "autotune=0 block=256 det_shared=0"; do
  args=""; for kv in $v; do args="$args --opt $kv"; done
  n=$(echo $v | tr ' =' '__')
  timeout -k 10 250 python3 tools/eval_loop.py --workload opf --supports 1000000 --products $args > $O/opf_$n.json 2>$O/opf_$n.err || echo fail $n
  python3 - <<PY
import json
j=json.loads(open("$O/opf_$n.json").read().strip().splitlines()[-1])
print("$v", {k:round(j["ms"][k]*1e3,1) for k in j["ms"]})
PY
done
