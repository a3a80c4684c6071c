#!/usr/bin/env python3
"""One bench-like process: ONE jac and ONE hess buffer (allocated like bench.py does), the default and the large-batch
code object timed into them (iem_time_kernels, 50 launches each).  Run several times: a census of the placement modes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from infiniteexamodels.jl_amd import transcribe, workloads
from infiniteexamodels.jl_amd.model import ExaModel
import bench
S = 1_000_000
core = transcribe.exa_core(workloads.quadrotor(S))
blob = core.to_blob()
a = ExaModel(core, device=0, blob=blob, options={"autotune": 0})
x, y = bench.eval_point(a.meta.nvar, a.meta.ncon, a.meta.x0, S, seed=0)
xd, yd = torch.tensor(x, device="cuda"), torch.tensor(y, device="cuda")
jac = torch.empty(a.meta.nnzj, dtype=torch.float64, device="cuda")
hess = torch.empty(a.meta.nnzh, dtype=torch.float64, device="cuda")
b = ExaModel(core, device=0, blob=blob, options={"autotune": 0, "lds_slots": 48})
out = []
for rnd in range(2):
    for name, m in (("default", a), ("lds48", b)):
        j, h = m.time_kernels(xd, yd, jac, hess, iters=50)
        out.append(f"{name} {j:.4f}/{h:.4f}")
print(" | ".join(out), f"| jac @0x{jac.data_ptr():x} hess @0x{hess.data_ptr():x}")
if os.environ.get("IEM_CENSUS_TAG"):
    t = ExaModel(core, device=0, blob=blob, options={"autotune": 0, "lds_slots": 48, "name_tag": 48})
    for name, mm in (("lds48 untagged", b), ("lds48 tagged", t)):
        j, h = mm.time_kernels(xd, yd, jac, hess, iters=50)
        print(f"{name}: {j:.4f}/{h:.4f} jit={any(k['jit'] for k in mm.kernels())}")
if os.environ.get("IEM_CENSUS_TUNE"):
    for it in (10, 50):
        j, h = b.time_kernels(xd, yd, jac, hess, iters=it)
        print(f"lds48 standalone, iters {it}: {j:.4f}/{h:.4f}")
    c = ExaModel(core, device=0, blob=blob)            # tuner on: its second code object is the same lds48 program
    print("iem_tune ->", c.tune(xd, yd, jac, hess))
    j, h = c.time_kernels(xd, yd, jac, hess, iters=50)
    print(f"tuned handle after iem_tune, iters 50: {j:.4f}/{h:.4f}")
