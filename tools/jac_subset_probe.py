#!/usr/bin/env python3
"""Which templates of the quadrotor Jacobian cost what?  Times jac_coord! for template subsets at
10^6 supports (ODE rows: S items each; finite-difference rows: S-1 items starting at support 2,
whose COO blocks sit 24 bytes off the 128-byte lines)."""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from infiniteexamodels.jl_amd import transcribe, workloads
from infiniteexamodels.jl_amd.core import T_CON
from infiniteexamodels.jl_amd.model import ExaModel
import bench

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
METHOD = sys.argv[2] if len(sys.argv) > 2 else "backward"     # forward: S-1 items starting at support 1 (aligned block starts)
im = workloads.quadrotor(S)
if METHOD != "backward":
    from infiniteexamodels.jl_amd import infinite as io
    im.set_derivative_method(im.groups[0].prefs[0], io.FiniteDifference(METHOD))
PAD = int(sys.argv[3]) if len(sys.argv) > 3 else 0     # extra point constraints AFTER the ODE rows: shifts the FD blocks
for i in range(PAD):
    im.constraint(im.infinite_variables[i % 9](0) == 0)
full = transcribe.exa_core(im)
print("derivative method:", METHOD)
cons = [t for t in full.templates if t.kind == T_CON]
first_fd = min(i for i, t in enumerate(full.templates) if t.kind == T_CON and len(t.items) == S - 1)
pos = {id(t): i for i, t in enumerate(full.templates)}
sets = {"fd + pad": lambda t: len(t.items) == S - 1 or (len(t.items) == 1 and pos[id(t)] > 9), "all": lambda t: True, "ode (S items)": lambda t: len(t.items) == S, "fd (S-1 items)": lambda t: len(t.items) == S - 1,
        "ode+point": lambda t: len(t.items) != S - 1}
for name, keep in sets.items():
    core = copy.copy(full)
    core.templates = [t for t in full.templates if t.kind != T_CON or keep(t)]
    core.ncon = sum(len(t.items) for t in core.templates if t.kind == T_CON)
    gm = ExaModel(core, device=0)
    x, y = bench.eval_point(gm.meta.nvar, gm.meta.ncon, gm.meta.x0, S)
    xd, yd = torch.tensor(x, device="cuda"), torch.tensor(y, device="cuda")
    j = torch.empty(gm.meta.nnzj, dtype=torch.float64, device="cuda")
    h = torch.empty(gm.meta.nnzh, dtype=torch.float64, device="cuda")
    ms_j, ms_h = gm.time_kernels(xd, yd, j, h, iters=100)
    k = {kk["kind"]: kk for kk in gm.kernels() if kk["grid"][0] > 1}
    bj = k["jac"]["alg_bytes_read"] + k["jac"]["alg_bytes_written"]
    bh = k["hess"]["alg_bytes_read"] + k["hess"]["alg_bytes_written"]
    print(f"{name:16s} nnzj {gm.meta.nnzj:9d} jac {ms_j:.4f} ms {bj / ms_j / 1e6:7.0f} GB/s | nnzh {gm.meta.nnzh:9d} hess {ms_h:.4f} ms {bh / ms_h / 1e6:7.0f} GB/s", flush=True)
    gm.close()
