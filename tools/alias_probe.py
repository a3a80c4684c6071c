"""In-process probe: does kernel time depend on the power of two dividing S, or on the output buffer?
(profiles/r01_size_vs_placement_4e6.txt) — three model sizes, the same three output buffer pairs."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from infiniteexamodels.jl_amd import transcribe, workloads
from infiniteexamodels.jl_amd.model import ExaModel
import bench
sizes = [4_000_000, 4_000_037, 4_194_304]
models = []
for S in sizes:
    gm = ExaModel(transcribe.exa_core(workloads.quadrotor(S)), device=0)
    x, y = bench.eval_point(gm.meta.nvar, gm.meta.ncon, gm.meta.x0, S)
    models.append((S, gm, torch.tensor(x, device="cuda"), torch.tensor(y, device="cuda")))
nj = max(m[1].meta.nnzj for m in models); nh = max(m[1].meta.nnzh for m in models)
bufs = [(torch.empty(nj, dtype=torch.float64, device="cuda"), torch.empty(nh, dtype=torch.float64, device="cuda")) for _ in range(3)]
for rnd in range(2):
    for S, gm, xd, yd in models:
        row = []
        for jb, hb in bufs:
            ms_j, ms_h = gm.time_kernels(xd, yd, jb[:gm.meta.nnzj], hb[:gm.meta.nnzh], iters=30)
            row.append(f"{ms_j:.4f}/{ms_h:.4f}")
        print(f"round {rnd} S={S:8d} jac/hess ms into 3 buffer pairs: " + "  ".join(row), flush=True)
