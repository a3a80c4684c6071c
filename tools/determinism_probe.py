#!/usr/bin/env python3
"""Ten calls of grad!/jtprod!/hprod! on a model: do the bits change?  (tools; the asserted cases are tests/test_gpu_determinism.py)
  python tools/determinism_probe.py quadrotor_oc3 20000"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from infiniteexamodels.jl_amd import transcribe, workloads
from infiniteexamodels.jl_amd.model import ExaModel

wl, S = sys.argv[1], int(sys.argv[2])
im = {"quadrotor_oc3": lambda: workloads.quadrotor(S, collocation=3), "hovercraft": lambda: workloads.hovercraft(S),
      "kinetic": lambda: workloads.kinetic_control(S), "pandemic": lambda: workloads.pandemic(S // 40 - 10, 40)}[wl]()
gm = ExaModel(transcribe.exa_core(im), device=0)
rng = np.random.default_rng(0)
x = torch.tensor(np.abs(gm.meta.x0 + 0.1 * rng.standard_normal(gm.meta.nvar)) + 0.05, device="cuda")
y = torch.tensor(rng.standard_normal(gm.meta.ncon), device="cuda")
v = torch.tensor(rng.standard_normal(gm.meta.nvar), device="cuda")
first, changed = None, set()
for it in range(20):
    out = (gm.grad(x).cpu().numpy().tobytes(), gm.jtprod(x, y).cpu().numpy().tobytes(), gm.hprod(x, y, v).cpu().numpy().tobytes())
    if first is None:
        first = out
    else:
        for a, b, what in zip(first, out, ("grad", "jtprod", "hprod")):
            if a != b:
                changed.add(what)
print(wl, S, "changed between calls:", sorted(changed) or "nothing")
