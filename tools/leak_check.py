#!/usr/bin/env python3
"""Create / evaluate / destroy handles in a loop: device memory must come back (second code object, gather plans, axis tables,
reduction buffers, mailboxes are all owned by the handle)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from infiniteexamodels.jl_amd import transcribe, workloads
from infiniteexamodels.jl_amd.model import ExaModel

torch.cuda.init()
for name, im in (("quadrotor 3e5 (two code objects)", workloads.quadrotor(300_000)), ("quadrotor OC3 5e4 (gather plan)", workloads.quadrotor(50_000, collocation=3)),
                 ("pandemic 600x40 (axis sums)", workloads.pandemic(590, 40))):
    core = transcribe.exa_core(im)
    blob = core.to_blob()
    free = []
    for it in range(25):
        gm = ExaModel(core, device=0, blob=blob)
        x = torch.tensor(np.abs(gm.meta.x0) + 0.1, device="cuda")
        y = torch.ones(gm.meta.ncon, dtype=torch.float64, device="cuda")
        gm.jac_coord(x); gm.hess_coord(x, y); gm.jtprod(x, y); gm.grad(x); gm.obj(x)
        torch.cuda.synchronize()
        gm.close()
        del x, y
        torch.cuda.empty_cache()
        free.append(torch.cuda.mem_get_info()[0])
    drift = (free[4] - free[-1]) / 2**20
    print(f"{name}: free after 5 cycles {free[4] / 2**20:.0f} MiB, after 25 cycles {free[-1] / 2**20:.0f} MiB, drift {drift:.1f} MiB")
    assert drift < 16, "device memory is not returned"
print("OK")
