#!/usr/bin/env python3
"""Probe: does the jac_coord!/hess_coord! kernel time depend on WHERE the output buffer sits?
(bimodal 86 / 106 us jac times were seen between otherwise identical runs)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from infiniteexamodels.jl_amd import transcribe, workloads
from infiniteexamodels.jl_amd.model import ExaModel
import bench

S = 1_000_000
core = transcribe.exa_core(workloads.quadrotor(S))
gm = ExaModel(core, device=0)
x, y = bench.eval_point(gm.meta.nvar, gm.meta.ncon, gm.meta.x0, S)
xd, yd = torch.tensor(x, device="cuda"), torch.tensor(y, device="cuda")
pad = 1 << 22
rows = []
for trial in range(3):
    filler = torch.empty((trial * 37 + 1) * 1_000_003, dtype=torch.float64, device="cuda")   # shifts later allocations
    jbuf = torch.empty(gm.meta.nnzj + pad, dtype=torch.float64, device="cuda")
    hbuf = torch.empty(gm.meta.nnzh + pad, dtype=torch.float64, device="cuda")
    for off in (0, 16, 512, 8192, 262144, 262144 + 16, 1 << 20):
        j, h = jbuf[off:off + gm.meta.nnzj], hbuf[off:off + gm.meta.nnzh]
        ms_j, ms_h = gm.time_kernels(xd, yd, j, h, iters=50)
        rows.append(dict(trial=trial, off=off, jac_addr=hex(j.data_ptr()), hess_addr=hex(h.data_ptr()), jac_ms=round(ms_j, 4), hess_ms=round(ms_h, 4)))
        print(rows[-1], flush=True)
    del filler, jbuf, hbuf
    torch.cuda.empty_cache()
