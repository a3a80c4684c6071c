#!/bin/bash
# SQ counters of the hub-mode KKT object at config 3 (20 x 20 blocks: the lane-per-row kernels)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04_kkt_pmc_hub}; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $O/sq -- python3 $R/tools/kkt_cabi_hub_profile.py --iters 1 > $O/sq.log 2>&1 || { tail -5 $O/sq.log; exit 1; }
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SMEM --output-format csv -d $O/sq2 -- python3 $R/tools/kkt_cabi_hub_profile.py --iters 1 > $O/sq2.log 2>&1 || { tail -5 $O/sq2.log; exit 1; }
python3 - $O <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
tot = {}
for d in ("sq", "sq2"):
    f = glob.glob(f"{O}/{d}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0]
        if k.startswith("kkt_"): acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, c in acc.items(): tot.setdefault(k, {}).update(c)
for k, c in sorted(tot.items()):
    w = c.get("SQ_WAVES", 1)
    print(k, f"waves {w:.3g}", " ".join(f"{n[3:]}/wave={v / w:.0f}" for n, v in sorted(c.items()) if n != "SQ_WAVES"))
PY
