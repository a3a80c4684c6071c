#!/usr/bin/env python3
"""Joins tools/placement_counters.py's per-window times with the rocprofv3 --pmc CSV of the SAME run:
`python3 tools/placement_summary.py <log with the JSON lines> <rocprof output dir> <label>` → one JSON object."""
import collections
import csv
import glob
import json
import os
import sys

log, d, label = sys.argv[1], sys.argv[2], sys.argv[3]
wins = [json.loads(l) for l in open(log) if l.startswith("{")]
per = wins[0]["launches_per_window"] if wins else 0
rows = collections.defaultdict(lambda: collections.defaultdict(list))   # kernel -> dispatch id -> [(counter, value)]
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"] in ("iem_jac_g0", "iem_hess_g0"):
            rows[r["Kernel_Name"]][int(r["Dispatch_Id"])].append((r["Counter_Name"], float(r["Counter_Value"])))
out = {"label": label, "windows": []}
for w in wins:
    e = dict(w)
    for kern, key in (("iem_jac_g0", "jac"), ("iem_hess_g0", "hess")):
        ids = sorted(rows[kern])
        mine = ids[w["window"] * per:(w["window"] + 1) * per]
        acc = collections.defaultdict(list)
        for i in mine:
            tot = collections.defaultdict(float)
            inst = collections.defaultdict(list)
            for c, v in rows[kern][i]:
                tot[c] += v          # a raw (non _sum) counter has one row per hardware instance
                inst[c].append(v)
            for c, v in tot.items():
                acc[c].append(v)
            for c, v in inst.items():
                if len(v) > 1:
                    acc[c + ":instances"].append(len(v))
                    acc[c + ":max_over_mean"].append(max(v) / (sum(v) / len(v)) if sum(v) else 0.0)
                    acc[c + ":min_over_mean"].append(min(v) / (sum(v) / len(v)) if sum(v) else 0.0)
        e[key + "_counters"] = {c: sum(v) / len(v) for c, v in acc.items()}
    out["windows"].append(e)
print(json.dumps(out, indent=1))
