#!/usr/bin/env python3
"""What does the halo exchange add to a shard's jac+hess step?  ONE process, TWO sharded handles (ranks 0 and 1 of a
world of 2) on cuda:0, each on a stream of its own, mailboxes wired in-process (same-pid path of iem_comm_connect) —
the closest a one-GPU box gets to two GPUs: the ranks' kernels really run concurrently (no inter-process time slicing),
only the links are missing.  Per variant: time per iteration (both ranks' steps enqueued back to back, one host thread).

  python tools/halo_overlap_probe.py [--supports 250000] [--iters 300]
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from infiniteexamodels.jl_amd import transcribe, workloads
from infiniteexamodels.jl_amd.model import ExaModel

ap = argparse.ArgumentParser()
ap.add_argument("--supports", type=int, default=250_000)
ap.add_argument("--iters", type=int, default=300)
ap.add_argument("--world", type=int, default=2)
args = ap.parse_args()

blob = transcribe.exa_core(workloads.quadrotor(args.supports)).to_blob()
W = args.world
gms = [ExaModel.sharded(blob, 1, r, W, device=0) for r in range(W)]
handles = b"".join(g.comm_export() for g in gms)
for g in gms:
    g.comm_connect(handles)
streams = [torch.cuda.Stream() for _ in range(W)]
st = []
for r, g in enumerate(gms):
    x = torch.tensor(g.meta.x0 + 0.1 * np.random.default_rng(r).standard_normal(g.meta.nvar), device="cuda")
    y = torch.tensor(np.random.default_rng(10 + r).standard_normal(g.meta.ncon), device="cuda")
    jac = torch.empty(g.meta.nnzj, dtype=torch.float64, device="cuda")
    hess = torch.empty(g.meta.nnzh, dtype=torch.float64, device="cuda")
    c = torch.empty(g.meta.ncon, dtype=torch.float64, device="cuda")
    st.append((x, y, jac, hess, c))
torch.cuda.synchronize()


def variant(fused, halo):
    steps = []
    for r, g in enumerate(gms):
        x, y, jac, hess, c = st[r]
        with torch.cuda.stream(streams[r]):
            if halo == "blocking":
                pair = g.raw_pair(x, y, jac, hess, fused=fused, halo=False)
                L, h, px = g._L, g._h, x.data_ptr()
                steps.append(lambda pair=pair, L=L, h=h, px=px: (L.iem_halo_exchange(h, px), pair()))
            else:
                steps.append(g.raw_pair(x, y, jac, hess, fused=fused, halo=(halo == "async")))
    return steps


def run(steps, n):
    for _ in range(20):
        for s in steps:
            s()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        for s in steps:
            s()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


out = {"supports": args.supports, "world": W, "own_n": [g.shard_info()["own_n"] for g in gms], "us_per_iteration": {}}
for fused in (False, True):
    for halo in ("none", "blocking", "async"):
        key = ("fused" if fused else "two calls") + " / halo " + halo
        ts = [run(variant(fused, halo), args.iters) for _ in range(3)]
        out["us_per_iteration"][key] = [round(t, 2) for t in ts]
        print(key, out["us_per_iteration"][key], flush=True)
# one rank alone (the other idle): the step itself
for fused in (False, True):
    s = variant(fused, "none")[:1]
    out["us_per_iteration"][("fused" if fused else "two calls") + " / rank 0 alone"] = [round(run(s, args.iters), 2) for _ in range(3)]
for g in gms:
    assert g.comm_status() == 0
print(json.dumps(out))
