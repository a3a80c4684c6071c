#!/usr/bin/env python3
"""COO→CSR assembly rate at the headline size (quadrotor, 1e6 supports)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from infiniteexamodels.jl_amd import transcribe, workloads
from infiniteexamodels.jl_amd.csr import CsrAssembler
from infiniteexamodels.jl_amd.model import ExaModel
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
gm = ExaModel(transcribe.exa_core(workloads.quadrotor(S)), device=0)
x = torch.tensor(gm.meta.x0 + 0.1 * np.random.default_rng(0).standard_normal(gm.meta.nvar), device="cuda")
y = torch.tensor(np.random.default_rng(1).standard_normal(gm.meta.ncon), device="cuda")
out = {}
for which in ("jac", "hess"):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    asm = CsrAssembler(gm, which)
    torch.cuda.synchronize(); plan_s = time.perf_counter() - t0
    coo = gm.jac_coord(x) if which == "jac" else gm.hess_coord(x, y)
    dst = torch.empty(asm.nnz, dtype=torch.float64, device="cuda")
    for _ in range(5): asm.values(coo, dst)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): asm.values(coo, dst)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    alg = 8 * (asm.n_coo + asm.nnz) + 8 * (asm.n_coo + asm.nnz + 1)   # values in/out + perm/seg indices
    out[which] = dict(n_coo=asm.n_coo, nnz_csr=asm.nnz, plan_s=plan_s, values_ms=ms, GBps=alg / (ms * 1e-3) / 1e9)
print(json.dumps(out))
