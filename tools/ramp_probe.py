#!/usr/bin/env python3
"""Does a 20-step timed region right after an idle gap run slower than in steady state?  One process, one buffer pair:
(idle 0.5 s, W warm-up pairs, 20 timed pairs) for W = 5, 50, 500, three times each."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from infiniteexamodels.jl_amd import transcribe, workloads
from infiniteexamodels.jl_amd.model import ExaModel
import bench
S = 1_000_000
core = transcribe.exa_core(workloads.quadrotor(S))
gm = ExaModel(core, device=0)
x, y = bench.eval_point(gm.meta.nvar, gm.meta.ncon, gm.meta.x0, S, seed=0)
xd, yd = torch.tensor(x, device="cuda"), torch.tensor(y, device="cuda")
jac = torch.empty(gm.meta.nnzj, dtype=torch.float64, device="cuda")
hess = torch.empty(gm.meta.nnzh, dtype=torch.float64, device="cuda")
step = gm.raw_pair(xd, yd, jac, hess, obj_weight=1.0)
for rep in range(3):
    row = []
    for W in (5, 50, 500):
        torch.cuda.synchronize(); time.sleep(0.5)
        for _ in range(W):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        row.append(f"W={W}: {(time.perf_counter() - t0) / 20 * 1e3:.5f}")
    print("rep", rep, " ".join(row), flush=True)
