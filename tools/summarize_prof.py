#!/usr/bin/env python3
"""Condenses a tools/profile_gpu.sh output directory into the JSON kept under profiles/:
`python tools/summarize_prof.py gpurun_out/prof profiles/rNN_rocprof_quadrotor_1e6.json`."""
import collections
import csv
import glob
import json
import os
import sys


def counters(d):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc, info = collections.defaultdict(list), {}
        for r in csv.DictReader(open(f)):
            if not r["Kernel_Name"].startswith("iem_"):
                continue
            acc[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
            info[r["Kernel_Name"]] = dict(vgpr=int(r["VGPR_Count"]), sgpr=int(r["SGPR_Count"]), lds=int(r["LDS_Block_Size"]),
                                          grid=int(r["Grid_Size"]), wg=int(r["Workgroup_Size"]))
        for (k, c), v in acc.items():
            out.setdefault(k, {"dispatch": info[k]})[c] = sum(v) / len(v)
    return out


def main():
    src, dst = sys.argv[1], sys.argv[2]
    stats = []
    for f in glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if r["Name"].startswith("iem_")]
        if rows:
            stats = [{k: r[k] for k in ("Name", "Calls", "AverageNs", "MinNs", "MaxNs", "StdDev", "Percentage")} for r in rows]
    pmc = {}
    for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
        if os.path.isdir(d):
            for k, c in counters(d).items():
                pmc.setdefault(k, {}).update(c)
    dur = {r["Name"]: float(r["AverageNs"]) * 1e-9 for r in stats}
    for k, c in pmc.items():
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            # MI355X_MICROARCH.md: gfx950 FETCH_SIZE tallies 128-B requests at 64 B → ×2; both are in KiB
            c["hbm_bytes_per_launch"] = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
        if "SQ_INSTS_VALU" in c and k in dur:
            clk = 2.4e9
            c["compute_guard"] = {
                "valu_insts_per_wave": c["SQ_INSTS_VALU"] / c["SQ_WAVES"],
                "valu_busy_frac_of_simd_cycles": c.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (1024 * dur[k] * clk),
                "fp64_tflops_upper_bound": c["SQ_INSTS_VALU"] * 64 * 2 / dur[k] / 1e12, "fp64_vector_peak_tflops": 78.6,
                "note": "every vector instruction counted as an FP64 FMA on 64 lanes; quad-cycles x4; 1024 SIMDs; 2.4 GHz assumed"}
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    commit = os.environ.get("IEM_COMMIT", "n/a")      # no .git on the GPU box: pass `IEM_COMMIT=$(git rev-parse --short HEAD)` through the runner
    json.dump({"csrc_fingerprint": bench.csrc_fingerprint(), "commit": commit, "command": "tools/profile_gpu.sh (rocprofv3 --kernel-trace --stats; each --pmc group in its own pass) on "
                          "python3 bench.py --no-cpu-baseline", "kernel_stats": stats, "pmc": pmc}, open(dst, "w"), indent=1)
    print(json.dumps({"kernel_stats": stats, "pmc": {k: {n: v for n, v in c.items() if n in ("hbm_bytes_per_launch", "compute_guard")}
                                                     for k, c in pmc.items()}}, indent=1))


if __name__ == "__main__":
    main()
