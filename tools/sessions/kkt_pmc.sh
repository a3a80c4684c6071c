#!/bin/bash
# SQ counters of the chain KKT kernels at 1e5 quadrotor supports: is kkt_eliminate bound by issue slots or by its dependency chain?
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04_kkt_pmc}; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $O/sq -- python3 $R/tools/kkt_chain_bench.py --iters 2 --cabi 0 > $O/sq.log 2>&1 || { tail -5 $O/sq.log; exit 1; }
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_MFMA --output-format csv -d $O/sq2 -- python3 $R/tools/kkt_chain_bench.py --iters 2 --cabi 0 > $O/sq2.log 2>&1 || { tail -5 $O/sq2.log; exit 1; }
python3 - $O <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
for d in ("sq", "sq2"):
    f = glob.glob(f"{O}/{d}/**/*counter_collection.csv", recursive=True)
    if not f: print("no csv in", d); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0]
        if not k.startswith("kkt_"): continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, c in acc.items():
        print(d, k, {n: f"{v:.4g}" for n, v in sorted(c.items())})
PY
