R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_qtrace; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 $R/tools/kkt_chain_bench.py --iters 3 --cabi 0 > $O/run.json 2> $O/run.err || { tail -3 $O/run.err; exit 1; }
python3 - $O <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/t/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0] for r in rows]
# last factorisation: the last run of 17 kkt_eliminate launches before solves
idx = [i for i, n in enumerate(names) if n == "kkt_eliminate"]
# group consecutive factor sequences: find last block of eliminate/update alternation
last = idx[-1]
start = last
while start > 0 and names[start - 1] in ("kkt_eliminate", "kkt_update"): start -= 1
seq = rows[start:last + 1]
el = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in seq if r["Kernel_Name"].startswith("kkt_eliminate")]
up = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in seq if r["Kernel_Name"].startswith("kkt_update")]
print("eliminate us:", " ".join(f"{d:.0f}" for d in el), "sum %.0f" % sum(el))
print("update us:", " ".join(f"{d:.0f}" for d in up), "sum %.0f" % sum(up))
print("span us %.0f" % ((int(seq[-1]["End_Timestamp"]) - int(seq[0]["Start_Timestamp"])) / 1e3))
PY
