#!/usr/bin/env python3
"""The reference's own benchmark ladder (ESCAPE34/run_cases_gpu.jl:90-102) on one MI355X:
collocation quadrotor and stochastic OPF at 1000…16000 supports, pandemic at the five (nt, nξ)
pairs — per-call device times of the five NLPModels calls, the eager loop and the same loop as a
replayed HIP graph.  The reference logs the solver's total "function evaluation" time for these
cases (ESCAPE34/utils.jl:7,23) and commits no values; this is the per-iteration cost that time is
made of.  Prints one JSON object per case and writes them all to --out."""
import argparse
import json
import os
import sys
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import eval_loop  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    cases = [("quadrotor_oc3", dict(supports=n)) for n in (1000, 2000, 4000, 8000, 16000)]
    cases += [("opf", dict(supports=n)) for n in (1000, 2000, 4000, 8000, 16000)]
    cases += [("pandemic", dict(nt=nt, nxi=nxi)) for nt, nxi in ((25, 4), (50, 4), (100, 4), (100, 8), (100, 128))]
    rows = []
    for wl, kw in cases:
        args = SimpleNamespace(workload=wl, nt=kw.get("nt", 0), nxi=kw.get("nxi", 0), supports=kw.get("supports", 0), iters=a.iters)
        r = eval_loop.measure(args)
        rows.append(r)
        print(json.dumps(r), flush=True)
    if a.out:
        json.dump(rows, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
