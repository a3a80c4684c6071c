#!/usr/bin/env python3
"""Placement by SELECTION: hold K candidate output buffers at once (a freed buffer's pages come straight back on the next
hipMalloc, so re-drawing one buffer finds nothing new — docs/lab_notebook.md), time the real call into each, keep the
fastest, free the rest.  Does the choice survive (same time afterwards, fused step included)?

  python tools/probes/pick_buffer_probe.py [--supports 1000000] [--candidates 8]
"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from infiniteexamodels.jl_amd import transcribe, workloads
from infiniteexamodels.jl_amd.model import ExaModel

ap = argparse.ArgumentParser()
ap.add_argument("--supports", type=int, default=1_000_000)
ap.add_argument("--candidates", type=int, default=8)
ap.add_argument("--iters", type=int, default=30)
args = ap.parse_args()
gm = ExaModel(transcribe.exa_core(workloads.quadrotor(args.supports)), device=0)
x = torch.tensor(gm.meta.x0 + 0.1 * np.random.default_rng(0).standard_normal(gm.meta.nvar), device="cuda")
y = torch.tensor(np.random.default_rng(1).standard_normal(gm.meta.ncon), device="cuda")


def timed(fn, iters=args.iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


new = lambda n: torch.empty(n, dtype=torch.float64, device="cuda")
# what a caller gets without looking: the first draw of each
j0, h0 = new(gm.meta.nnzj), new(gm.meta.nnzh)
base = dict(jac=timed(lambda: gm.jac_coord(x, j0)), hess=timed(lambda: gm.hess_coord(x, y, h0)), pair=timed(lambda: gm.jac_hess_coord(x, y, j0, h0)))
out = {"supports": args.supports, "first_draw_ms": base}
print("first draw", {k: round(v, 4) for k, v in base.items()}, flush=True)
cj = [j0] + [new(gm.meta.nnzj) for _ in range(args.candidates - 1)]
tj = [timed(lambda b=b: gm.jac_coord(x, b)) for b in cj]
ch = [h0] + [new(gm.meta.nnzh) for _ in range(args.candidates - 1)]
th = [timed(lambda b=b: gm.hess_coord(x, y, b)) for b in ch]
print("jac candidates ", [round(t, 4) for t in tj], flush=True)
print("hess candidates", [round(t, 4) for t in th], flush=True)
bj, bh = cj[int(np.argmin(tj))], ch[int(np.argmin(th))]
del cj, ch, j0, h0
torch.cuda.empty_cache()          # the rejected candidates go back to the driver
again = dict(jac=timed(lambda: gm.jac_coord(x, bj)), hess=timed(lambda: gm.hess_coord(x, y, bh)), pair=timed(lambda: gm.jac_hess_coord(x, y, bj, bh), 100))
print("chosen, after freeing the others", {k: round(v, 4) for k, v in again.items()}, flush=True)
out.update(jac_candidates_ms=tj, hess_candidates_ms=th, chosen_ms=again)
print(json.dumps(out))
