#!/usr/bin/env python3
"""LagrangeNewtonSolver on the quadrotor at several sizes: iterations, steps, shifts.  python tools/probes/newton_sizes.py [sizes...]"""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
warnings.simplefilter("ignore")
from infiniteexamodels.jl_amd import workloads
from infiniteexamodels.jl_amd.backend import ExaTranscriptionBackend
from infiniteexamodels.jl_amd.model import MI355XBackend
from infiniteexamodels.jl_amd.contrib.newton import LagrangeNewtonSolver
sizes = [int(a) for a in sys.argv[1:]] or [500, 1000, 2000, 5000, 20000, 100000]
for n in sizes:
    im = workloads.quadrotor(n, backend=ExaTranscriptionBackend(LagrangeNewtonSolver(tol=1e-8, max_iter=60), backend=MI355XBackend()))
    r = im.optimize()
    print(n, r.status, r.iterations, r.kkt_residual, flush=True)
    for h in r.history[:14] + r.history[-2:]:
        print("   ", {k: (float(f"{v:.4g}") if isinstance(v, float) else v) for k, v in h.items() if k in ("iter", "kkt_residual", "obj", "step", "delta_w", "factorisations", "merit_weight")})
