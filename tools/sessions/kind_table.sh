#!/bin/bash
# per-kind kernel durations (rocprofv3 --kernel-trace --stats) of the evaluation loop + matrix-free products at the headline size
# and at BASELINE config 3: the numbers behind DESIGN.md's per-kind roofline table
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-kind_table}; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/q1e6 -- python3 $R/tools/eval_loop.py --workload quadrotor --supports 1000000 --products --iters 50 > $O/q1e6.json 2> $O/q1e6.err || { tail -3 $O/q1e6.err; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pandemic -- python3 $R/tools/eval_loop.py --workload pandemic --products --iters 50 > $O/pandemic.json 2> $O/pandemic.err || { tail -3 $O/pandemic.err; exit 1; }
python3 - $O <<'PY'
import csv, glob, json, os, sys
O = sys.argv[1]
for tag in ("q1e6", "pandemic"):
    j = json.load(open(os.path.join(O, tag + ".json")))
    f = glob.glob(os.path.join(O, tag, "**", "*kernel_stats.csv"), recursive=True)[0]
    dur = {r["Name"]: (float(r["AverageNs"]), int(r["Calls"])) for r in csv.DictReader(open(f)) if r["Name"].startswith("iem_")}
    print(j["workload"])
    kinds = {"obj": "obj", "grad": "grad", "cons": "cons", "jac_coord": "jac", "hess_coord": "hess", "jprod": "jprod", "jtprod": "jtprod", "hprod": "hprod"}
    for call, kind in kinds.items():
        names = [n for n in dur if n.startswith("iem_" + kind + "_")]
        if not names or call not in j["alg_bytes"]:
            continue
        us = sum(dur[n][0] for n in names) / 1e3
        b = j["alg_bytes"][call]
        print(f"  {call:11s} {'+'.join(names):28s} {b / 1e6:8.1f} MB  {us:7.1f} us  {b / us / 1e3 / 8000:.3f} of 8 TB/s")
PY
