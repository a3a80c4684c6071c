#!/bin/bash
# per-kernel time of the hub-mode KKT object at config 3: factor and solve in separate rocprofv3 runs
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04_hub_prof}; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for what in factor solve; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$what -- python3 $R/tools/kkt_cabi_hub_profile.py --iters 5 --what $what > $O/$what.json 2> $O/$what.err || { tail -5 $O/$what.err; exit 1; }
  cat $O/$what.json
  f=$(ls -t $(find $O/$what -name "*kernel_stats.csv") | head -1)
  python3 - $f $what <<'PY' | tee $O/${what}_kernels.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"# {sys.argv[2]}: kernels by total time (whole process: create + warm-up + 6 timed calls)")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:22]:
    print(f"{float(r['TotalDurationNs']) / 1e6:9.3f} ms  {int(r['Calls']):6d} calls  {float(r['AverageNs']) / 1e3:9.1f} us avg  {r['Name'][:90]}")
PY
done
